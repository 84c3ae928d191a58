echo "== gauge loads non-temporal (default)"; python tools/hopsplit_ab.py 2>&1 | grep -E "L=(12|16|20|24)" | head -16
echo "== gauge loads temporal"; TMLQCD_HIP_LIB=$GRAFT_REPO_ROOT/tmlqcd_amd/lib_ab/libtmlqcd_hip_t.so python tools/hopsplit_ab.py 2>&1 | grep -E "L=(12|16|20|24)" | head -16
