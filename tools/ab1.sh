# one bench.py line per variant: ab1 NAME "bench args" ENV=...
ab1() { name=$1; args=$2; shift 2; env "$@" python bench.py --no-cpu-baseline --no-extras $args 2>&1 | tail -1 | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read()); print('$name', 'fps %.1f'%d['value'], {a:round(b*1000,1) for a,b in d['stage_ms_single_stream'].items()}, 'err', d['error_flags'], {k:v for k,v in d.get('parity',{}).items() if k in ('psnr_db_min','keys_bit_exact','ids_match','error')})
except Exception as e: print('$name FAILED', e)
"; }
