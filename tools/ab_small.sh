set -e
for b in 1024 2048; do
  echo "batch $b small"; FBS_BR_SMALL=1 python3 bench.py --batch $b --steps 10 --warmup 3 --cpu-sample 0 --no-secure | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
done
echo "batch 1024 standard"; python3 bench.py --batch 1024 --steps 10 --warmup 3 --cpu-sample 0 --no-secure | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])"
