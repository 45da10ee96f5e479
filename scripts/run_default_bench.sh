mkdir -p gpurun_out/r6_final
s=$(date +%s)
python bench.py > gpurun_out/r6_final/bench_default_stdout.txt 2> gpurun_out/r6_final/bench_default_stderr.txt
echo rc=$? seconds=$(( $(date +%s) - s ))
wc -c gpurun_out/r6_final/bench_default_stdout.txt
python - <<'P'
import json
d=json.loads(open("gpurun_out/r6_final/bench_default_stdout.txt").read().strip().splitlines()[-1])
print(d["value"], d["steps"], d["warmup"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["reorder"])
print(json.load(open("bench_full.json"))["phase_seconds"])
P
