cd $GRAFT_REPO_ROOT
VIPRS_HIP_LIB=build/libviprs_hip_cvtplain.so timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -x -q -k "quantised" 2>&1 | tail -5
VIPRS_HIP_LIB=build/libviprs_hip_cvtplain.so timeout 300 python bench.py --no-secondary --cpu-seconds 0 --model grid --ld-dtype int8 | cut -c1-200
