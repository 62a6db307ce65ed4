# python3 bench.py over batch sizes (P1024, reduced noise): bash tools/batch_sweep.sh [sizes...] > gpurun_out/batch_sweep.txt
SIZES=${@:-64 128 256 384 512 768 1024 1280 1536 2048 3072 4096 8192 16384}
for b in $SIZES; do
  python3 bench.py --batch $b --steps 6 --warmup 2 --cpu-sample 0 --no-secure 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']
print('batch %6d  %8.0f FBS/s  %8.3f ms/step  br %7.3f ms (%s)  ks %.3f ms (%s) ok=%s' % ($b, d['value'], d['ms_per_step'], r['avg_launch_ms'], r['kernel'], r['keyswitch_avg_launch_ms'], r['keyswitch_kernel'], d['decrypt_ok']))"
done
