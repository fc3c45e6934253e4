export SHAPES="4096,160,960;4096,320,1280;4096,512,256;4096,512,1024;4096,576,160;4096,960,160;4096,960,320;4096,1024,75;4096,1024,512;4096,1280,512;16384,64,384;16384,96,576;16384,192,64;16384,256,128;16384,256,512;16384,352,256;16384,384,64;16384,384,96;16384,512,75;16384,512,256;16384,576,96;65536,32,192;65536,128,256;65536,144,32;65536,160,128;65536,192,32;65536,256,75;65536,256,128"
mkdir -p gpurun_out/r2
for bm in def 64 128; do for bn in def 64 128; do
  if [ $bm = def ] && [ $bn != def ]; then continue; fi
  if [ $bm != def ] && [ $bn = def ]; then continue; fi
  unset PQ_G2_BM PQ_G2_BN
  if [ $bm != def ]; then export PQ_G2_BM=$bm PQ_G2_BN=$bn; fi
  echo "== bm $bm bn $bn"
  timeout -k 10 200 python scripts/gemm_bench.py fwd,dgrad 30 2>&1 | grep "^M="
done; done
