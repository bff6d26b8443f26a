#!/bin/bash
# round 3, GPU call 29: conjugate-pair discriminator filters (k_2fsk_ff / k_disc_2fsk): parity, goldens, C1 step
set -u
export TMPDIR=/tmp
O=gpurun_out/r03ac
rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_lifecycle.py tests/test_golden.py tests/test_golden_extra.py tests/test_gpu_modem_facade.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
tail -n 5 $O/pytest.log
for m in "" "--no-overlap"; do
python bench.py --config c1 --steps 15 --warmup 3 --no-extra --check $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c1 $m', d['ms_per_step'], d['value'], d['roofline']['kernel'], d['roofline']['kernel_ms'], d.get('parity_check',{}).get('status'))"
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kprof -o alone_c1 -- python tools/kprof.py 18 16384 262144 1000000 3 > $O/kprof.log 2>&1
for f in $(find $O/kprof -name '*_results.db' | sort); do python tools/prof_summary.py $f alone_c1; done
find $O -name '*.db' -delete; find $O -name '*.csv' -size +1M -delete
