for R in 64 128 256 512; do
  TAG=res$R bash tools/gpu.sh bench c3 --steps 4 --warmup 1 --hook seed_reserve=$R || exit 1
done
