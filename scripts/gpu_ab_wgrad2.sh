cd "${GRAFT_REPO_ROOT:-/root/repo}"
MICRO=0 TAIL=8 bash scripts/gpu_r05.sh r05_e "tests/test_gpu_parity.py" -
for round in 1 2; do for wm in 1 2; do
NL_WGRAD2_MODE=$wm timeout 300 python bench.py --no-cpu-baseline --no-api-path --no-large-map --no-settings --no-pmc 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('wgrad2 mode $wm  ms/step %.4f  sustained %.4f  decoder %.4f  dW2 %.4f (frac %.3f)' % (d['ms_per_step'], d['steady_state']['ms_per_step'], r['avg_launch_ms'], r['second_kernel']['avg_launch_ms'], r['second_kernel']['frac']))"
done; done
