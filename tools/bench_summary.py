import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    print(f"{f}: value={d['value']:.1f} ms/step={d['ms_per_step']:.0f} gemmTF={r['achieved']:.1f} share={r['kernel_time_share']:.2f} stepTF={d['step_tflops']:.1f} kernels={d['kernel_ms']}")
