#!/bin/bash
# SQ counters of the bf16 conv kernels (one pass; rocprofv3 --pmc in its own run, as the guide prescribes)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pmc1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES ${PMC_EXTRA:-SQ_BUSY_CYCLES} SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmc1 -o p --output-format csv -- python3 "$ROOT/tools/bench_encoder.py" 32 1000 2 Cnn10 ${1:-bf16} > /dev/null 2>&1
python3 - <<'PY'
import csv, collections, glob
f = glob.glob('/tmp/pmc1/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'conv_igemm' in k or 'conv_wgrad' in k or 'conv_wino' in k:
        short = ('wino_wgrad' if 'wino_wgrad' in k else 'wino' if 'wino' in k else 'igemm' if 'igemm' in k else 'wgrad') + ('<128>' if 'Li128' in k or '<128' in k else '<64>' if 'Li64' in k or '<64' in k else '')
        agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    wc = m.get('SQ_WAVE_CYCLES', 1)
    print(k, 'launches', len(v['SQ_WAVE_CYCLES']))
    for c in sorted(m):
        print('   %-22s %14.0f  %6.1f %% of wave cycles' % (c, m[c], 100 * m[c] / wc))
PY
