import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    km = d.get("kernel_ms") or d.get("kernel_ms_sampled_launches_only")
    print(f"{f}: value={d['value']:.1f} ms/step={d['ms_per_step']:.0f} gemmTF={r['achieved']:.1f} k256TF={r.get('achieved_k256_launches_only', 0):.1f} stepTF={d['step_tflops']:.1f} kernels={km}")
