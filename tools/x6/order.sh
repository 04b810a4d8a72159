#!/bin/bash
# prints the instruction order of the main loop of one kernel (mangled-name substring) from /tmp/x6/x6.s
#   M mfma, r ds_read, W ds_write, G global load, v VALU, | waitcnt, BAR barrier
name=$1
awk -v n="$name" '$0 ~ "^_Z.*"n".*:" {f=1} f{print} /^\.Lfunc_end/{if(f)exit}' /tmp/x6/x6.s > /tmp/x6/one.s
# the inner loop = the last "Inner Loop Header" block up to its backward branch
awk '/Inner Loop Header/{f=1; buf=""} f{buf=buf"\n"$0} /s_cbranch/{if(f){last=buf}; }END{print last}' /tmp/x6/one.s | awk '{print $1}' | awk '
/v_mfma/{printf "M "; next} /ds_read/{printf "r "; next} /ds_write/{printf "W "; next} /global_load|buffer_load/{printf "G "; next} /^v_/{printf "v "; next} /s_waitcnt/{printf "| "; next} /s_barrier/{printf "BAR "; next} /s_nop/{printf "n "; next}'
echo
