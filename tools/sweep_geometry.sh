for c in 2 3; do for v in 2 1 3; do for m in 1 2 3 4 6 8; do
g=$((256*m))
echo -n "cfg $c variant $v grid $g: "
python bench.py --config $c --variant $v --grid $g --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e9,1), round(d['ms_per_step'],4), round(r['achieved']), r['kernel'], round(r['avg_launch_ms'],4))"
done; done; done
