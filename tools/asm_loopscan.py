import re,sys
s=open(sys.argv[1]).read()
pat=sys.argv[2]
lines=s.split('\n')
starts=[i for i,l in enumerate(lines) if re.match(r'_Z\w+:', l)]
starts.append(len(lines))
for a,b in zip(starts,starts[1:]):
    name=lines[a].split(':')[0]
    if pat not in name: continue
    seg_all=lines[a:b]
    labels={}
    for i,l in enumerate(seg_all):
        m=re.match(r'(\.LBB\d+_\d+):',l)
        if m: labels[m.group(1)]=i
    best=None
    for i,l in enumerate(seg_all):
        m=re.search(r's_cbranch_\w+ (\.LBB\d+_\d+)',l)
        if m and m.group(1) in labels and labels[m.group(1)]<i:
            seg=seg_all[labels[m.group(1)]:i]
            nm=sum('v_mfma' in x for x in seg)
            if best is None or nm>best[0]: best=(nm,seg)
    if not best: continue
    nm,seg=best
    body='\n'.join(seg_all)
    print(name[:70], 'loop: mfma',nm,'readlane',sum('v_readlane' in x for x in seg),'writelane',sum('v_writelane' in x for x in seg),'valu',sum(x.strip().startswith('v_') and 'mfma' not in x for x in seg), '| fn readlane', body.count('v_readlane'), 'writelane', body.count('v_writelane'))
