import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line)
    print("fps %.1f"%d["value"], {k:round(v,3) for k,v in d["stage_ms"].items()}, "D=%.2fM V=%.2fM err=%d"%(d["visible_splats"]["tile_pairs"]/1e6, d["visible_splats"]["sorted"]/1e6, d["error_flags"]))
