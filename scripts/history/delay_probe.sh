for d in 0 2 4 8 16 32; do echo "nap $d"; GOMEL_PERSIST_NAP=$d timeout -k 10 100 python scripts/persist_sweep.py 2>&1 | grep "frames  342\|frames  128\|frames 1024\|frames  512"; done
