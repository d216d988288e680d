# SQ instruction / busy counters of a workload's kernels (four rocprofv3 --pmc passes, kernel trace only): instructions per
# wavefront by class, VALU / matrix-pipe busy cycles, cycles spent in s_waitcnt.  Run on the MI355X box from the repo root:
#   WL=gnn EXTRA='--periods 4' bash tools/pmc_sq_probe.sh        (WL: any bench.py workload)
cd ${GRAFT_REPO_ROOT:-.}
export WL=${WL:-gnn}
export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  d=gpurun_out/pmc_${WL:-gnn}/$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 bench.py --workload ${WL:-gnn} --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing $EXTRA > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob,collections,re
tot=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/pmc_'+__import__('os').environ.get('WL','gnn')+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=re.sub(r"\(anonymous namespace\)::","",r["Kernel_Name"]); k=re.sub(r"^void ","",k).split("(")[0]
        if not k.startswith(("mlp3","segment","gnn_","env_","small_","closed_")): continue
        k=k+"|"+r["Grid_Size"]
        tot[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[k][r["Counter_Name"]]+=1
for k in sorted(tot):
    v={c:tot[k][c]/cnt[k][c] for c in tot[k]}
    w=v.get("SQ_WAVES",0) or 1
    print(k, "launches",cnt[k].get("SQ_WAVES"))
    print("   waves %.0f  per wave: VALU %.0f SALU %.0f VMEM_RD %.0f VMEM_WR %.0f LDS %.0f MFMA %.0f" % (w, v.get("SQ_INSTS_VALU",0)/w, v.get("SQ_INSTS_SALU",0)/w, v.get("SQ_INSTS_VMEM_RD",0)/w, v.get("SQ_INSTS_VMEM_WR",0)/w, v.get("SQ_INSTS_LDS",0)/w, v.get("SQ_INSTS_MFMA",0)/w))
    print("   " + " ".join(f"{c}={v[c]:.3g}" for c in sorted(v) if not c.startswith("SQ_INSTS")))
PY
