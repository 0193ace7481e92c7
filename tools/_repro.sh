run() { echo "== $3 $1 | tmin=$2"; RAYLIB_LIB=$PWD/software-raytracing_amd/libraylib$3.so FUZZ_MODES_JSON="$1" FUZZ_TMIN_OVERRIDE="$2" FUZZ_ONLY=667 FUZZ_TMIN=1 timeout 40 python tools/gpu_fuzz.py 2000 20261004 2>&1 | grep -E "^fuzz|MISMATCH|stuck" | sort | uniq -c | sort -rn | head -4; }
run '[{"RAYLIB_POOL": "0"}]' -0.0001 _wd
run '[{"RAYLIB_POOL": "0"}, {"RAYLIB_POOL": "0", "RAYLIB_LEAF_LIST": "0"}, {"RAYLIB_POOL": "2"}]' -0.0001 ""
run '[{"RAYLIB_POOL": "0"}, {"RAYLIB_POOL": "0", "RAYLIB_LEAF_LIST": "0"}, {"RAYLIB_POOL": "2"}]' -0.05 ""
