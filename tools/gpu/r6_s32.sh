cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r6s32; mkdir -p $O
line() { name=$1; shift; ( timeout 900 python bench.py "$@" ) > $O/$name.out 2>&1; grep '^{' $O/$name.out | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('  ', round(d['ms_per_step'],4), 'ms/step')
print('   ' + ' | '.join('%s %.1f/%.0f' % (f['family'][:9], f['us_per_step'], f['launches_per_step']) for f in r.get('families', [])))
for k in (r.get('kernels') or [])[:12]: print('     %-90s %5.1f x %6.1f' % (k['kernel'][:90], k['launches_per_step'], k['us_per_step']))
" | tee -a $O/lines_summary.txt; }
G="--no_cpu_baseline --steady_steps 0 --precondition 64 --config configs/dcn_v2_criteo.config --dense_dtype bf16"
echo f32_tail | tee -a $O/lines_summary.txt; line a $G
echo bf16_grouped | tee -a $O/lines_summary.txt; EASYREC_AMD_BF16_WGRAD_F32=0 line b $G
