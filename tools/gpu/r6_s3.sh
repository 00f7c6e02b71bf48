#!/bin/bash
# round 6 session 3: DIN's first attention layer on the generated [q, h, q - h, q * h] operand (er_din_gemm_*): kernel and
# model tests, same-box A/B of BASELINE config 4
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s3; mkdir -p $O
timeout 900 python -m pytest tests/test_fused_epilogues_gpu.py -q --timeout 300 -x -k "din" 2>&1 | tail -15 | tee $O/tests_new.txt
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_kv_embedding.py tests/test_deepfm_gpu.py -q --timeout 300 -m gpu -k "din or DIN or hash_table_sequence or trajectory or neighbouring" 2>&1 | tail -8 | tee $O/tests_models.txt
line() { name=$1; shift; ( timeout 600 python bench.py "$@" ) > $O/$name.out 2>&1; echo "$name exit $?"; grep '^{' $O/$name.out | tail -1 >> $O/bench_lines.jsonl; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}; p=d.get('parity_full_size') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step', round(d['value']), '| parity', p.get('max_rel_loss_diff'), p.get('ok'))
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:22]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
Q="--config configs/din_taobao_10m.config --steady_steps 0 --precondition 128 --cpu_seconds 2"
echo din10m_generated | tee -a $O/lines_summary.txt; line din10m_generated $Q
echo din10m_built | tee -a $O/lines_summary.txt; EASYREC_AMD_DIN_FUSED=0 line din10m_built $Q --no_cpu_baseline
ls $O; du -sh $O
