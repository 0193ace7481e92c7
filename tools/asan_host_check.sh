#!/bin/bash
# AddressSanitizer + UBSan over the host side of the library (no GPU involved; GPU sanitizers are unavailable on the pool).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/libraylib_asan.so
SRC="$ROOT/software-raytracing_amd/csrc"
g++ -std=c++17 -O1 -g -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -DRAYLIB_EXPORTS=1 \
    -I"$ROOT/include" -I"$SRC" "$SRC"/rl_abi.cc "$SRC"/rl_scene.cc "$SRC"/rl_bvh.cc "$SRC"/rl_cull.cc "$SRC"/rl_obj_loader.cc "$SRC"/rl_image_io.cc "$SRC"/rl_jpeg.cc "$SRC"/rl_log.cc \
    "$ROOT/tools/nodevice_stub.cc" -o "$OUT" -lz -lpthread
cd "$ROOT"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LD_PRELOAD="$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)" RAYLIB_LIB="$OUT" \
    python -m pytest tests/test_host_logic.py tests/test_image_codecs.py -x -q
# mutation fuzz of the image decoders under the same sanitizers (a decoder may refuse a file; it must not misbehave)
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:allocator_may_return_null=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LD_PRELOAD="$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)" RAYLIB_LIB="$OUT" \
    python tools/fuzz_codecs.py ${FUZZ_FILES:-6000} ${FUZZ_SEED:-1} | tail -1
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:allocator_may_return_null=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
LD_PRELOAD="$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)" RAYLIB_LIB="$OUT" \
    python tools/fuzz_obj.py ${FUZZ_OBJ_FILES:-2000} ${FUZZ_SEED:-1} | tail -1
