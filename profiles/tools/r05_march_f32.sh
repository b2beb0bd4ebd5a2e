# the march micro-benchmark with a double and with a float state: loads one plane ahead (k_march2) and two planes ahead (k_march3)
cd profiles/micro
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o march_stencil march_stencil.hip
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -DREAL=float -o march_stencil_f32 march_stencil.hip
echo "== double state"; ./march_stencil
echo "== float state"; ./march_stencil_f32
echo "== double state again"; ./march_stencil
