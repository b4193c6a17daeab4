import sys, json
for l in sys.stdin:
    if not l.startswith("{"):
        print(l[:300].rstrip()); continue
    try: d = json.loads(l)
    except Exception as e:
        print("unparsable", l[:200]); continue
    print(d["n_gpus"], d["value"], d["ms_per_step"], d["config"]["recurrence"], d["config"].get("recurrence_trial_ms_per_step"), d.get("extra_legs_error"), d.get("extra_legs_s"))
    s = d.get("strong", {})
    print("   strong:", {k: (v.get("ms_per_iteration", v) if isinstance(v, dict) else v) for k, v in s.items() if k in ("single_reduction", "pipelined", "single_reduction+p2p", "recurrence")})
    p = d.get("configs3_p2", {})
    print("   p2:", p.get("recurrence"), p.get("recurrence_trial_ms_per_step"), p.get("max_abs_error_vs_exact_profile"))
