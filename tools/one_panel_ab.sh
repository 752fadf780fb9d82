cd $GRAFT_REPO_ROOT
for np in 1024 2048; do for n in 200 224 240 256; do for sw in "" 0; do
  SL_GP4_ONE_PANEL=$sw python bench.py --config C2 --num-points $np --n-gp $n --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$np^2 n=$n ONE_PANEL=$sw: ms_per_step %.4f kernel_ms %.4f %s' % (d['ms_per_step'], r['kernel_ms'], r['kernel'][:70]))"
done; done; done
for n in 200 224 256; do for sw in "" 0; do
  SL_GP4_ONE_PANEL=$sw python bench.py --config C4 --num-points 48 --n-gp $n --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('48^4 n=$n ONE_PANEL=$sw: ms_per_step %.4f kernel_ms %.4f %s' % (d['ms_per_step'], r['kernel_ms'], r['kernel'][:70]))"
done; done
