for B in 768 1024 768 1024; do echo "RT06_BLOCK=$B"; RT06_BLOCK=$B RT06_DEBUG=1 python tools/perf_variants.py 3 2>&1 | grep -v amdgpu | tail -2; done
