"""Short summary of a bench.py line (file argument or stdin)."""
import json, sys
d = json.loads((open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"NIZK 2^20: {d['ms_per_step']} ms = {d['value']/1e6:.1f} M/s; stages {d.get('stage_ms')}")
print(f"dominant {r['kernel_name']}: {r['avg_launch_ms']} ms, frac {r['frac']}, alu {r['alu']['achieved']}/{r['alu']['peak']} (isa frac {r['alu'].get('frac_isa')}); classes {r['chosen_from']}")
print("field_mul peak", d["field_mul"]["peak"], {k: (v["ms_per_proof"], v["frac"]) for k, v in d["field_mul"]["classes"].items()})
for k, v in (d.get("sweep") or {}).items():
    if isinstance(v, dict) and "ms_per_proof" in v: print(" sweep", k, v["ms_per_proof"], "ms", v.get("stage_ms"), "ok" if v.get("equals_oracle_digest") else "DIGEST?")
s = d.get("snark") or {}
print("snark", {k: s.get(k) for k in ("ms_per_proof", "value", "verify_ms", "encode_ms", "equals_oracle_digest", "stage_ms")})
print("in_flight", (d.get("in_flight") or {}).get("value"), "verify_ms", d.get("verify_ms"), "prepare_device_ms", d.get("prepare_device_ms"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
print("e2e", (d.get("spzk_e2e") or {}).get("wall_ms"), "floor", d.get("hip_process_floor_ms"), "equals digest", d.get("equals_oracle_digest"), d.get("proof_sha256", "")[:12])
