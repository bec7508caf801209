#!/bin/bash
# copies the record run's outputs (gpurun_out/r04final, written by tools/gpu_final_r04.sh on the GPU box) into profiles/
F=gpurun_out/r04final; P=profiles
cp $F/bench_c3d4.json $P/r04_bench_c3d4_n1.json; cp $F/bench_c3d4_three_kernel.json $P/r04_bench_c3d4_n1_three_kernel.json
cp $F/bench_c3d10.json $P/r04_bench_c3d10_n1.json; cp $F/bench_c3d10_node_order_vectors.json $P/r04_bench_c3d10_n1_node_order_vectors.json
cp $F/bench_forcecomm_persistent.json $P/r04_bench_forcecomm_persistent_across_ranks.json; cp $F/bench_forcecomm_rccl.json $P/r04_bench_forcecomm_rccl_loop.json
cp $F/bench_shm_n2.json $P/r04_bench_shm_n2.json; cp $F/bench_shm_n4.json $P/r04_bench_shm_n4.json
cp $F/kernel_stats_c3d4.txt $P/r04_kernel_stats_bench_c3d4.txt; cp $F/kernel_stats_c3d10.txt $P/r04_kernel_stats_bench_c3d10.txt
cp $F/pmc_fetch_c3d4.txt $P/r04_pmc_fetch_size_c3d4.txt; cp $F/pmc_write_c3d4.txt $P/r04_pmc_write_size_c3d4.txt
cp $F/pmc_fetch_c3d10.txt $P/r04_pmc_fetch_size_c3d10.txt; cp $F/pmc_write_c3d10.txt $P/r04_pmc_write_size_c3d10.txt
cp $F/pmc_spmv_c3d10.txt $P/r04_pmc_spmv_c3d10_storage_vs_node_order.txt; cp $F/persist_inband.txt $P/r04_persist_inband.txt
cp $F/ab_order_c3d10.txt $P/r04_ab_order_c3d10.txt; cp $F/ab_order_c3d4_8m.txt $P/r04_ab_order_c3d4_8m.txt
cp $F/spmv_knobs.txt $P/r04_spmv_knobs.txt; cp $F/ab_fused.txt $P/r04_ab_fused_update.txt; cp $F/ab_footprint.txt $P/r04_ab_footprint_product.txt
cp $F/microbench.txt $P/r04_microbench_k12_c3d10.txt; cp $F/spmv_traffic.json $P/spmv_traffic.json
