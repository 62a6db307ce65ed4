# python3 bench.py over batch sizes (P1024, reduced noise): bash tools/batch_sweep.sh > gpurun_out/batch_sweep.txt
for b in 64 256 512 768 1024 1536 2048 3072 4096 8192 16384; do
  python3 bench.py --batch $b --steps 6 --warmup 2 --cpu-sample 0 --no-secure 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('batch %6d  %8.0f FBS/s  %8.3f ms/step  br %7.3f ms (%s)  ks %.3f ms' % ($b, d['value'], d['ms_per_step'], r['avg_launch_ms'], r['kernel'], r['keyswitch_avg_launch_ms']))"
done
