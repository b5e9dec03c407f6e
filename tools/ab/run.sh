for v in "$@"; do cp tools/ab/$v.so nerf_amd/libnerf_hip.so && echo "== $v" && timeout -k 10 100 python tools/time_fine.py 2>&1 | grep us || exit 1; done
