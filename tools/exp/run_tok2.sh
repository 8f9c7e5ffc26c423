cd /root/repo
timeout 200 python tools/exp/tok_occ.py 2>&1 | tail -1
for lib in $LIBS; do SVX_EXP_LIB=$(pwd)/svision_amd/$lib.so timeout 200 python tools/exp/tok_occ.py 2>&1 | tail -1; done
timeout 900 python -m pytest tests/test_gpu_inflate.py -x -q -m gpu --timeout 300 2>&1 | tail -3
