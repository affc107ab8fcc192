set -e
mkdir -p gpurun_out/r3u
: > gpurun_out/r3u/ab_lib.txt
for rep in 1 2 3; do
  for v in B3 B4; do
    cp _ab/lib$v.so descriptools_amd/libdescriptools_hip.so
    echo "lib $v" >> gpurun_out/r3u/ab_lib.txt
    timeout -k 10 200 python3 tools/ab_key.py 5 0 16384 8 2>&1 | grep "key 5" | head -2 >> gpurun_out/r3u/ab_lib.txt
  done
done
cp _ab/libB4.so descriptools_amd/libdescriptools_hip.so
