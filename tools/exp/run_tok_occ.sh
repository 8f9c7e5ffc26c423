cd /root/repo
for lds in 0 4000 12000 26000; do SVX_TOK_LDS=$lds timeout 200 python tools/exp/tok_occ.py 2>&1 | tail -1; done
for lib in $LIBS; do SVX_EXP_LIB=$(pwd)/svision_amd/$lib.so timeout 200 python tools/exp/tok_occ.py 2>&1 | tail -1; done
