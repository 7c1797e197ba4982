set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for m in 0 1 2 3; do
  export HX_XCD_ROWMAP=$m
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rm_$m -- python3 $R/bench.py --no-sweep --no-cpu-baseline --steps 3000 --warmup 300 --settle-s 0.5 --reps 1 > $R/gpurun_out/rm_$m.log 2>&1
  f=$(ls $R/gpurun_out/rm_$m/*/*kernel_stats.csv | head -1)
  python3 - "$f" $m <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
out=[]
tot=0
w={'act_fused_kernel<1, false, true':1,'wgrad':1.5,'fwd_l2_kernel<64, true, true':1,'fwd_l2_kernel<32':1,'bwd_l2_kernel<0':1,'fwd_l2_kernel<64, true, false':0.5,'bwd_l2_kernel<2':0.5,'bwd_l2_kernel<1':0.5}
for r in rows[:9]:
    n=r["Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    for k,v in w.items():
        if n.startswith(k) or (k=='wgrad' and n.startswith('wgrad')):
            tot+=v*float(r["AverageNs"])/1e3
            out.append("%s %.2f"%(n[:28], float(r["AverageNs"])/1e3)); break
print("rowmap", sys.argv[2], "sum %.2f |"%tot, " | ".join(out))
PY
  rm -rf $R/gpurun_out/rm_$m $R/gpurun_out/rm_$m.log
done
done
