"""One-line digest of a bench.py JSON line (tools/gpu.sh prints it after every bench task)."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:                                 # noqa: BLE001
        print(f, "UNREADABLE:", e, open(f).read()[-400:] if __import__("os").path.exists(f) else "")
        continue
    r, x, c = d.get("roofline") or {}, d.get("extras") or {}, d.get("cpu_baseline") or {}
    print("%s: %.1f %s  %.3f ms/step %s  n_gpus %d  host issue %.2f ms  roofline %.4f (kernels %.2f ms)%s%s%s" % (
        f, d["value"], d["unit"], d["ms_per_step"], d.get("step_ms"), d["n_gpus"], d["config"].get("host_issue_ms_per_step", d["config"].get("host_launch_ms_per_step", 0.0)),
        r.get("frac") or 0.0, r.get("kernel_ms_per_step") or 0.0,
        "  dropin %.3f ms" % x["dropin"]["ms_per_step"] if "dropin" in x else "",
        "  B4 %.3f ms" % x["per_gpu_batch_4"]["ms_per_step"] if "per_gpu_batch_4" in x else "",
        "  cpu %.2f/s x%d" % (c["value"], c["cores"]) if c else ""))
    if r.get("per_kind_ms"):
        print("    ", r["per_kind_ms"])
    if r.get("dominant"):
        dm = r["dominant"]
        print("     dominant %s: %.2f ms, %.0f TF/s = %.3f;  wgrad %.2f ms, %.0f TF/s = %.3f" % (
            dm["kernel"].split(" ")[0], dm["ms_per_step"], dm["achieved"], dm["frac"], dm["second"]["ms_per_step"],
            dm["second"]["achieved"], dm["second"]["frac"]))
