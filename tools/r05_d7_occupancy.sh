for sh in 64,16 16,2 16,3 32,8; do for o in 3 2 1; do echo -n "wg/cu $o: "; NA_SHAPE=$sh PPNET_D7_WG_PER_CU=$o timeout -k 10 100 python tools/na_timing.py 2>&1 | grep side; done; done
