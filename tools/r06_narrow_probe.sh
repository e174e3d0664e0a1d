#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
timeout 1500 python tools/ab_sites.py --sites conv_narrow,state_lrelu --rounds 1 --steps 5 p0=lab p1=lab,CRFP_NARROW_PROBE=1 p2=lab,CRFP_NARROW_PROBE=2 p4=lab,CRFP_NARROW_PROBE=4 p5=lab,CRFP_NARROW_PROBE=5 p7=lab,CRFP_NARROW_PROBE=7 > gpurun_out/r06_narrow_probe.txt 2>&1
cat gpurun_out/r06_narrow_probe.txt
