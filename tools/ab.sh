# A/B of env knobs / variant libraries: bench.py --inflight 1 and 3, one line each
run() { name=$1; shift; echo "== $name"; env "$@" python bench.py --no-cpu-baseline --no-extras --inflight 1 2>&1 | tail -1 > gpurun_out/ab_${name}_if1.json; env "$@" python bench.py --no-cpu-baseline --no-extras 2>&1 | tail -1 > gpurun_out/ab_${name}_if3.json;
python - <<PY
import json
for k in ("if1","if3"):
    try:
        d=json.load(open("gpurun_out/ab_${name}_%s.json"%k)); s=d["stage_ms_single_stream"]
        print("${name}", k, "fps %.1f"%d["value"], "stages", {a:round(b*1000,1) for a,b in s.items()}, "err", d["error_flags"], "parity", {k:v for k,v in d.get("parity",{}).items() if k in ("psnr_db_min","keys_bit_exact","ids_match","error")})
    except Exception as e: print("${name}", k, "FAILED", e, open("gpurun_out/ab_${name}_%s.json"%k).read()[-400:])
PY
}
