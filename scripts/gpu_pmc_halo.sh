#!/bin/bash
# SQ stall breakdown of the halo conv (and whatever else the microbenchmark launches): one --pmc pass, 8 SQ counters
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
mkdir -p gpurun_out
cd /tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -d $R/gpurun_out/pmc_halo -o pmc --output-format csv -- python $R/${RQ_CMD:-scripts/conv_halo_bench.py} > $R/gpurun_out/pmc_halo.log 2>&1
echo "exit $?"
cd $R
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('gpurun_out/pmc_halo/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:70]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0))[:8]:
    wc = c.get('SQ_WAVE_CYCLES', 1) or 1
    print(k)
    print('   wait_any %.2f  wait_inst_any %.2f (lds %.2f)  active_inst %.2f  of wave cycles;  mfma_busy/busy_cycles %.3f  valu insts %.3g' % (
        c.get('SQ_WAIT_ANY', 0) / wc, c.get('SQ_WAIT_INST_ANY', 0) / wc, c.get('SQ_WAIT_INST_LDS', 0) / wc, c.get('SQ_ACTIVE_INST_ANY', 0) / wc,
        c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(c.get('SQ_BUSY_CYCLES', 1), 1), c.get('SQ_INSTS_VALU', 0)))
PY
rm -rf gpurun_out/pmc_halo
