#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side code (GPU sanitizers are not
# available on this pool): the oracle's renderer and the product's host helpers (scene reader with
# good, truncated and missing files, seed stream, camera basis).  Run from the repository root.
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd)
T=/tmp/rt_san; mkdir -p $T
python3 - <<PY
import sys; sys.path.insert(0, "$R")
from raytracing_simple_amd import host, scenes
sph, o, t = scenes.demo_plus(16)
host.write_scene("$T/a.scn", sph, o, t)
open("$T/bad1.scn", "w").write("camera 1 2 3\n")
open("$T/bad2.scn", "w").write(open("$T/a.scn").read()[:200])
PY
gcc -std=c11 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -mfma -ffp-contract=off -I$R/oracle \
    $R/tools/sanitize/oracle_main.c $R/oracle/rt_oracle.c -o $T/oracle_san -lm -lpthread
$T/oracle_san
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -ffp-contract=off -I$R/include \
    $R/tools/sanitize/host_main.cpp $R/raytracing_simple_amd/csrc/rt_host.cpp -o $T/host_san
$T/host_san
# the .scn reader on 1000 mutated files (deletions, junk tokens, non-finite numbers, truncation)
mkdir -p $T/fz
python3 - <<PY
import random
base = open("$T/a.scn", "rb").read()
random.seed(1)
toks = [b"nan", b"inf", b"-inf", b"1e999", b"-1e999", b"0x10", b"", b" ", b"\n", b"\x00", b"size", b"sphere", b"camera",
        b"99999999999999999999", b"-1", b"1.5e-45", b"\t\t", b"9" * 400]
for i in range(1000):
    b = bytearray(base)
    for _ in range(random.randint(1, 6)):
        k, pos = random.random(), random.randrange(len(b) + 1)
        if k < 0.3 and b: del b[pos:pos + random.randint(1, 40)]
        elif k < 0.6: b[pos:pos] = random.choice(toks)
        elif k < 0.8 and b: b[min(pos, len(b) - 1)] = random.randrange(256)
        else: b = b[:pos]
    open("$T/fz/f%d.scn" % i, "wb").write(bytes(b))
PY
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -ffp-contract=off -I$R/include \
    $R/tools/sanitize/reader_fuzz.cpp $R/raytracing_simple_amd/csrc/rt_host.cpp -o $T/reader_fuzz
$T/reader_fuzz
echo "sanitizers: clean"
