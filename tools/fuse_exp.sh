export CRFP_HIP_LIB=$PWD/crfp_amd/libcrfp_hip_lab.so
for p in 0 16 18 2; do echo "probe $p: $(CRFP_DCN_FUSE_PROBE=$p python tools/site_table.py 2>&1 | grep -E 'fused')"; done
