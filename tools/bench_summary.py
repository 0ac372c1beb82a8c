import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]
    iso = r.get("isolated_single_stream_pass") or {}
    km = iso.get("kernel_ms") or d.get("kernel_ms") or d.get("kernel_ms_estimated_from_sampled_launches")
    print(f"{f}: value={d['value']:.1f} ms/step={d['ms_per_step']:.0f} k256TF={r['achieved']:.1f} "
          f"isoK256={iso.get('achieved', 0):.1f} isoAll={iso.get('achieved_all_zgemm_launches', 0):.1f} "
          f"stepTF={d['step_tflops']:.1f} kernels={km}")
