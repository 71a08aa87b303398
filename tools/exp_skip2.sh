so=/tmp/librgbdvo_variant.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -Wno-unused-result -DRVO_SKIP_EXPERIMENT -o $so slam_amd/csrc/rvo.hip || exit 1
F=pyr_down,fast_grid,build_keypoints; R=init_rotation,match,match_finish; P=refine,refine_finish,consistency; T=clique,clique_walk,gauss_newton,track_ids
run() { RVO_SKIP=$2 RVO_LIB=$so RVO_BENCH_NOPROF=1 timeout -k 10 150 python bench.py --no-cpu --no-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('only $1:', round(d['ms_per_step'],4))" || exit 1; }
run R $F,$P,$T; run P $F,$R,$T; run T $F,$R,$P; run RP $F,$T; run RT $F,$P; run PT $F,$R; run RPT $F; run nothing $F,$R,$P,$T; run F $R,$P,$T; run FR $P,$T; run FP $R,$T; run FT $R,$P
