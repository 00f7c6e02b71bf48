#!/bin/bash
# round 2, call b: the whole -m gpu suite (no -x), bench default + A/B of the rolling flush / absorbed regime
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02b; mkdir -p $O
( time timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 ) > $O/pytest.log 2>&1
( time timeout 600 python bench.py ) > $O/bench.log 2>&1
for v in "EASYREC_AMD_FLUSH_WINDOWS=0" "EASYREC_AMD_FLUSH_WINDOWS=0 EASYREC_AMD_ABSORB=0" "EASYREC_AMD_FLUSH_WINDOWS=64" "EASYREC_AMD_FLUSH_WINDOWS=1024"; do
  echo "== $v" >> $O/ab.log
  env $v timeout 600 python bench.py --no_cpu_baseline --steps 200 --steady_steps 1024 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d.get('steady_state',{}); print(round(d['ms_per_step'],4), {k:round(v,4) for k,v in s.items() if isinstance(v,float)})" >> $O/ab.log 2>&1
done
tail -5 $O/pytest.log; tail -c 2500 $O/bench.log; cat $O/ab.log
