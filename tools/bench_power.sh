#!/bin/bash
# Package power and clocks while the config-2 / 4 / 5 benches run (rocm-smi samples every 0.25 s beside `bench.py --steps 4`): is the whole batch at the
# power limit, or only its dense launches?   bash tools/bench_power.sh > gpurun_out/r06_bench_power.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for C in 2 4 5; do
  ( while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/bp_smi_c$C.txt &
  SMI=$!
  python bench.py --config $C --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-second-mode --no-box-probe 2>/dev/null > gpurun_out/bp_bench_c$C.json
  kill $SMI; wait $SMI 2>/dev/null
  python - <<PY
import json, re, statistics as st
d = json.loads(open("gpurun_out/bp_bench_c$C.json").read().strip().splitlines()[-1])
pw, ck = [], []
for ln in open("gpurun_out/bp_smi_c$C.txt"):
    m = re.search(r"Power \(W\): ([0-9.]+)", ln) or re.search(r"Power[^:]*: ([0-9.]+)", ln)
    c = re.search(r"sclk[^(]*\((\d+)Mhz\)", ln)
    if m and c:
        pw.append(float(m.group(1))); ck.append(int(c.group(1)))
# the samples of the timed region: the busy tail of the run (setup draws < 400 W)
busy = [(p, c) for p, c in zip(pw, ck) if p > 600]
print(f"config $C: {d['value']:.3f} images/s, {d['ms_per_step']:.0f} ms per batch; rocm-smi samples {len(pw)}, of them above 600 W: {len(busy)}")
if busy:
    P, K = [b[0] for b in busy], [b[1] for b in busy]
    q = lambda v, f: sorted(v)[min(len(v) - 1, int(f * len(v)))]
    print(f"   power W: mean {st.mean(P):.0f}  p10 {q(P, .1):.0f}  median {q(P, .5):.0f}  p90 {q(P, .9):.0f}  max {max(P):.0f}")
    print(f"   sclk MHz: mean {st.mean(K):.0f}  p10 {q(K, .1)}  median {q(K, .5)}  p90 {q(K, .9)}  max {max(K)}")
PY
done
echo "--- raw samples (config 2, every fourth)"; awk 'NR % 4 == 0' gpurun_out/bp_smi_c2.txt | head -60
