cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
CMD="python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-profile --no-multi-stream"
for c in "VALUBusy" "MemUnitBusy" "MemUnitStalled" "SALUBusy" "LDSBankConflict" "L2CacheHit"; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/busy/$c -- $CMD > /dev/null 2> $ROOT/gpurun_out/busy/$c.err || echo "fail $c"
done
cd $ROOT
python3 - <<'PY'
import csv,glob,collections,re
res=collections.defaultdict(dict)
for d in glob.glob('gpurun_out/busy/*/'):
    c=d.rstrip('/').split('/')[-1]
    for f in glob.glob(d+'*/*counter_collection.csv'):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r['Counter_Name']==c:
                name=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void crfp::","").replace("crfp::","")
                acc[name].append(float(r['Counter_Value']))
        for k,v in acc.items(): res[k][c]=sum(v)/len(v)
cols=["VALUBusy","MemUnitBusy","MemUnitStalled","SALUBusy","LDSBankConflict","L2CacheHit"]
print(f"{'kernel':40s}"+"".join(f"{c:>16s}" for c in cols))
for k in ["conv3x3_split_kernel<1, 1, 2>","dcn_g8_pipe_kernel","conv3x3_narrow_kernel<1, 0>","conv3x3_narrow_kernel<2, 0>","conv3x3_narrow_kernel<3, 0>","dcn3_kernel","flow_warp_p4_kernel","flow_warp_p4_dual_kernel<8, 6>","hr_prep_kernel"]:
    if k in res: print(f"{k[:40]:40s}"+"".join(f"{res[k].get(c,float('nan')):16.1f}" for c in cols))
PY
rm -f gpurun_out/busy/*/*/*kernel_trace.csv
