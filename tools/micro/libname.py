import torch
dev=torch.device("cuda:0")
for (M,N,K) in [(32768,512,512),(32768,256,256),(32768,1024,512),(8192,8192,1024)]:
    A=torch.randn(M,K,device=dev); B=torch.randn(N,K,device=dev)
    for _ in range(3): torch.mm(A,B.t())
torch.cuda.synchronize()
