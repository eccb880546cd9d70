# round 3, GPU call 5: exact mode after the staging / half-wave / raw-record changes; PMC comparison exact vs approx;
# hunt for the round-2 multi-stream regression (count atomics?)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c5
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_exact_mode.py -x -q -m gpu > $O/pytest_exact.txt 2>&1; echo "exact rc=$?"; tail -5 $O/pytest_exact.txt
timeout 600 python tools/exact_probe.py --quick > $O/exact_probe.txt 2>&1; cat $O/exact_probe.txt
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $GRAFT_REPO_ROOT/$O/pmc_sq -o pmc -- python $GRAFT_REPO_ROOT/tools/exact_pmc.py 0.9 > $GRAFT_REPO_ROOT/$O/pmc_sq.txt 2>&1 )
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SMEM -d $GRAFT_REPO_ROOT/$O/pmc_mfma -o pmc -- python $GRAFT_REPO_ROOT/tools/exact_pmc.py 0.9 > $GRAFT_REPO_ROOT/$O/pmc_mfma.txt 2>&1 )
python - <<'PY'
import glob, sqlite3
for db in sorted(glob.glob("gpurun_out/r03c5/pmc_*/**/*.db", recursive=True)):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                       "where kernel_name like '%score_%' group by kernel_name, counter_name order by kernel_name").fetchall()
    print("==", db)
    for k, c, n, v, d in rows:
        print("%-70s %-26s %5d %16.1f %10.2f us" % (k.replace("(anonymous namespace)::", "")[:70], c, n, v, d / 1e3))
PY
find $O -name '*.db' -size +1M -delete
for i in 1 2; do
  timeout 300 python bench.py --approx --no-cpu-baseline --no-parity --regions 9 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/ab_approx_$i.json
  PVNET_SCORE_ATOMIC=0 timeout 300 python bench.py --approx --no-cpu-baseline --no-parity --regions 9 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/ab_approx_noatomic_$i.json
  ( cd _ab/r01 && timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > ../../$O/ab_r01_$i.json )
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r03c5/ab_*.json")):
    try:
        j = json.loads(open(f).read())
        print(f.split("/")[-1], round(j["value"]), j["ms_per_step"], (j.get("single_stream") or {}).get("value"), (j.get("regions") or {}).get("spread"))
    except Exception as e:
        print(f, "ERR", e)
PY
