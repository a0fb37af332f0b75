mkdir -p gpurun_out
python bench.py --config C3 --no-dense --no-cpu-baseline --timings 5 --steps 3 > gpurun_out/c3_t_graph.json 2> gpurun_out/c3_t_graph.err
BLOCKCOPY_GRAPH_TRAIN=0 python bench.py --config C3 --no-dense --no-cpu-baseline --timings 5 --steps 3 > gpurun_out/c3_t_eager.json 2> gpurun_out/c3_t_eager.err
grep -v "Warning\|warn\|amdgpu.ids" gpurun_out/c3_t_graph.err | tail -40
echo ======
grep -v "Warning\|warn\|amdgpu.ids" gpurun_out/c3_t_eager.err | tail -40
