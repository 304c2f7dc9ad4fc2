#!/bin/bash
# builds csrc/host_model.cpp + tools/fuzz_loaders.cpp with ASan + UBSan (CPU only) and runs the mutation fuzzer on the golden files
set -e
R=$(cd $(dirname $0)/.. && pwd); O=${TMPDIR:-/tmp}/mgs_fuzz; mkdir -p $O
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off \
    $R/tools/fuzz_loaders.cpp $R/vk_gaussian_splatting_amd/csrc/host_model.cpp -I$R/include -lz -lpthread -o $O/fuzz_loaders
N=${1:-3000}
export ASAN_OPTIONS=allocator_may_return_null=1:max_allocation_size_mb=8192 UBSAN_OPTIONS=halt_on_error=1
for f in ingest_ply_sh3.ply ingest_ply_sh0.ply ingest_ply_ascii.ply ingest_ply_be.ply vkrepro/scene.ply; do [ -f $R/tests/golden/$f ] && $O/fuzz_loaders $R/tests/golden/$f $N $O/m.ply; done
for f in ingest_spz_sh3.spz ingest_spz_sh1.spz ingest_spz_sh0.spz; do [ -f $R/tests/golden/$f ] && $O/fuzz_loaders $R/tests/golden/$f $N $O/m.spz; done
# .splat (antimatter15 layout, 32 bytes per splat: position 3xf32, scale 3xf32, rgba 4xu8, rotation 4xu8): a generated seed file
python3 - "$O/seed.splat" <<'PY'
import sys, numpy as np
rng = np.random.default_rng(1)
n = 500
rec = np.zeros(n, dtype=[("p", "<f4", 3), ("s", "<f4", 3), ("c", "u1", 4), ("r", "u1", 4)])
rec["p"] = rng.standard_normal((n, 3)); rec["s"] = np.exp(rng.standard_normal((n, 3)) - 4)
rec["c"] = rng.integers(0, 256, (n, 4)); rec["r"] = rng.integers(0, 256, (n, 4))
rec.tofile(sys.argv[1])
PY
$O/fuzz_loaders $O/seed.splat $N $O/m.splat
echo done
