"""Copy the round-2 measurements worth judging from gpurun_out/ (scratch) into profiles/ (tracked), as compact
summaries: per-kernel counter averages instead of raw counter CSVs, kernel-stats CSVs as they are, bench lines and
tuning logs as they are.  Idempotent; missing sources are skipped with a note."""
import collections
import csv
import glob
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out")
DST = os.path.join(ROOT, "profiles")


def copy(src, dst):
    s = os.path.join(SRC, src)
    if not os.path.exists(s):
        print("skip (absent):", src)
        return
    shutil.copyfile(s, os.path.join(DST, dst))
    print("copied", src, "->", dst)


def counters(src_dir, dst, keep=("conv_tile", "wgrad", "kmap", "k_sp_", "k_rs_", "conv_gather"), header=""):
    """Average every counter per kernel over the launches of each pass directory under src_dir."""
    d = os.path.join(SRC, src_dir)
    if not os.path.isdir(d):
        print("skip (absent):", src_dir)
        return
    lines = [header] if header else []
    for sub in sorted(os.listdir(d)):
        files = glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True)
        for f in files:
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            dur = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "")
                agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur[k].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            lines.append(f"== pass {sub}")
            for k, cs in agg.items():
                if not any(t in k for t in keep):
                    continue
                n = len(next(iter(cs.values())))
                avg = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
                lines.append(f"  {k[:110]}\n      launches={n} avg_ns={sum(dur[k]) / len(dur[k]):.0f} {avg}")
    with open(os.path.join(DST, dst), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("summarised", src_dir, "->", dst)


def stats(src_glob, dst):
    files = glob.glob(os.path.join(SRC, src_glob), recursive=True)
    if not files:
        print("skip (absent):", src_glob)
        return
    shutil.copyfile(files[0], os.path.join(DST, dst))
    print("copied", os.path.relpath(files[0], SRC), "->", dst)


def main():
    os.makedirs(DST, exist_ok=True)
    # fp32 tile kernels: tuning of the LDS-DMA family against the round-1 kernel, its PMC breakdown, HBM traffic
    for i, tag in enumerate(["r02b", "r02c", "r02c2", "r02c3"]):
        copy(f"{tag}/tune_dma.log", f"r02_tune_conv_dma_{i + 1}.log")
    counters("r02_pmc_dma", "r02_pmc_conv_dma.log",
             header="# rocprofv3 --pmc passes over scripts/prof_conv.py (config-2 conv 64->128, 834914 pairs); "
                    "SQ_* cycle counters are in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES = 32 x N_mfma")
    counters("r02g_traffic", "r02_pmc_traffic_tile_order.log",
             header="# FETCH_SIZE / WRITE_SIZE (KB, as reported: the gfx950 fetch correction x2 is applied in "
                    "DESIGN.md), TCC hit/miss; v8 = round-1 kernel, dma = LDS-DMA kernel <128,2,1>; rows / spatial = "
                    "tile order")
    # kernel-map build
    for wl in ("conv3d", "conv4d"):
        for lds in (0, 1):
            stats(f"r02f_kmap/{wl}_{lds}/**/*kernel_stats.csv", f"r02_rocprof_kernel_stats_kmap_{wl}_{'lds' if lds else 'flat'}.csv")
    # bench lines
    copy("r02a/bench_n2.json", "r02_bench_conv_n2_selfspawn_1gpu.json")
    copy("r02a/bench_unet_n2.json", "r02_bench_minkunet34c_n2_1gpu_ddp.json")
    copy("r02a/ddp_example.log", "r02_ddp_example_2ranks_1gpu.log")
    copy("r02i/bench.json", "r02_bench_conv_lpt.json")
    copy("r02i/bench4d.json", "r02_bench_conv4d_lpt.json")
    copy("r02i/bench_sparse.json", "r02_bench_conv_sparse_lpt.json")
    copy("r02g2/bench_v8_rows.json", "r02_bench_conv_v8_row_tiles.json")
    copy("r02g2/bench_v8_spatial.json", "r02_bench_conv_v8_spatial_tiles.json")
    copy("r02g2/bench_dma_rows.json", "r02_bench_conv_dma_row_tiles.json")
    copy("r02g2/bench_dma_spatial.json", "r02_bench_conv_dma_spatial_tiles.json")
    copy("r02j/unet_bf16_0.json", "r02_bench_minkunet34c_bf16_plan.json")
    copy("r02j/unet_bf16_1.json", "r02_bench_minkunet34c_bf16_gather_v1.json")
    copy("r02j/unet_bf16_graph.json", "r02_bench_minkunet34c_bf16_gather_v1_graph.json")
    copy("r02j/bench_bf16_0.json", "r02_bench_conv_bf16_plan.json")
    copy("r02j/bench_bf16_1.json", "r02_bench_conv_bf16_gather_v1.json")
    copy("r02k/bench_bf16_1.json", "r02_bench_conv_bf16_gather_v2_ring.json")
    copy("r02j/bench4d_bf16_0.json", "r02_bench_conv4d_bf16_plan.json")
    copy("r02j/bench4d_bf16_1.json", "r02_bench_conv4d_bf16_gather_v1.json")
    copy("config3_parity.log", "r02_config3_parity.log")
    # fp32 on the bf16 matrix pipe (k_conv_tile_f32x3)
    copy("tune_split1.log", "r02_tune_split_policy.log")
    copy("r02o/phase.log", "r02_phase_timing_f32x3_two_barriers.log")
    copy("r02p/phase.log", "r02_phase_timing_f32x3_pingpong.log")
    counters("r02r", "r02_pmc_traffic_f32x3.log", keep=("conv_tile", "wgrad"),
             header="# FETCH_SIZE / WRITE_SIZE (KB as reported; fetch x2 per the gfx950 correction), TCC hit / miss of the "
                    "config-2 launches with the round-2 default kernels (scripts/gpu_r02r.sh)")
    copy("r02l/bench_1.json", "r02_bench_conv_f32x3_first.json")
    copy("r02l/bench_0.json", "r02_bench_conv_f32_mfma_same_session.json")
    copy("r02r/bench.json", "r02_bench_conv_f32x3_pingpong.json")
    copy("r02r/bench4d.json", "r02_bench_conv4d_auto_policy.json")
    copy("r02r/unet_f32.json", "r02_bench_minkunet34c_f32_auto_policy.json")
    copy("r02r/unet_bf16.json", "r02_bench_minkunet34c_bf16_stage_pad.json")
    copy("r02r/pytest_gpu.log", "r02_pytest_gpu_full.log")
    copy("tune_wgrad_x3.log", "r02_tune_wgrad_f32x3.log")
    copy("pmc_wgrad_order.log", "r02_pmc_wgrad_range_order.log")
    copy("r02v/bench.json", "r02_bench_conv_f32x3_wgrad_split.json")
    copy("r02v/unet_f32.json", "r02_bench_minkunet34c_f32_wgrad_split.json")
    # closing session of the round (scripts/gpu_final_r02.sh)
    for src, dst in (("bench.json", "r02_bench_final.json"), ("bench_bf16.json", "r02_bench_final_bf16.json"),
                     ("bench_sparse.json", "r02_bench_final_sparse.json"), ("bench_conv4d.json", "r02_bench_final_conv4d.json"),
                     ("bench_conv4d_bf16.json", "r02_bench_final_conv4d_bf16.json"),
                     ("bench_n2.json", "r02_bench_final_n2_selfspawn_1gpu.json"),
                     ("unet_bf16.json", "r02_bench_final_minkunet34c_bf16.json"),
                     ("unet_f32.json", "r02_bench_final_minkunet34c_f32.json"),
                     ("unet_bf16_graph.json", "r02_bench_final_minkunet34c_bf16_graph.json"),
                     ("kernel_stats_bench.csv", "r02_rocprof_kernel_stats_final.csv"),
                     ("pytest_gpu.log", "r02_pytest_gpu_final.log"), ("smoke.log", "r02_smoke_final.log")):
        copy("r02_final8/" + src, dst)
    copy("r02_final8/unet_bf16_fresh.json", "r02_bench_final_minkunet34c_bf16_fresh_scenes.json")
    copy("r02_final8/unet_bf16_pipelined.json", "r02_bench_final_minkunet34c_bf16_pipelined_scenes.json")
    copy("bf16_batch_fusion.log", "r02_bench_bf16_batch_fusion.log")
    copy("r02_final3/bench.json", "r02_bench_before_wave_specialisation.json")
    copy("r02_final6/bench.json", "r02_bench_four_multipliers_best_box.json")
    copy("ablation_f32x3_ws.log", "r02_ablation_conv_f32x3_ws.log")
    copy("r02ws3/bench_v0.json", "r02_bench_conv_f32x3_ws.json")
    copy("r02ws3/unet_f32_v0.json", "r02_bench_minkunet34c_f32_ws.json")
    counters("r02_pmc_final", "r02_pmc_traffic_final.log", keep=("conv_tile", "wgrad"),
             header="# FETCH_SIZE / WRITE_SIZE (KB as reported; fetch x2 per the gfx950 correction), TCC hit / miss of the "
                    "config-2 launches of the kernels the round ends on (scripts/gpu_pmc_final.sh)")
    copy("r02ws5/bench_v0.json", "r02_bench_conv_f32x3_ws_eight_multipliers.json")
    copy("r02ws5/bench_v31.json", "r02_bench_conv_f32x3_ws_four_multipliers.json")
    copy("r02ws5/unet_f32_v0.json", "r02_bench_minkunet34c_f32_ws_eight_multipliers.json")
    # tile order of the matrix-bound fp32 kernels (end of the round): row tiles vs supercell order (position-space
    # maps, or the spatial index on flat-table maps = the default) vs Z-order argsort
    hdr = ("# FETCH_SIZE / WRITE_SIZE (KB as reported; fetch x2 per the gfx950 correction), TCC hit / miss of the "
           "config-2 launches (scripts/gpu_pmc_final.sh); ")
    counters("r02_pmc_tile_order_final", "r02_pmc_traffic_tile_order_final.log", keep=("conv_tile", "wgrad"),
             header=hdr + "DEFAULT: flat-table map, tiles in the supercell order of the target map")
    counters("r02_pmc_spatial", "r02_pmc_traffic_tile_order_position_space_maps.log", keep=("conv_tile", "wgrad"),
             header=hdr + "ME_AMD_SPATIAL_MAPS=1 ME_AMD_TILE_ORDER=spatial (position-space map, pair lists in position order)")
    counters("r02_pmc_zorder", "r02_pmc_traffic_tile_order_zorder.log", keep=("conv_tile", "wgrad"),
             header=hdr + "ME_AMD_SPATIAL_TILES=1 (argsort of the Z-order keys)")
    for src, dst in (("r02_exp8/c2_rows.json", "r02_bench_tile_order_rows.json"),
                     ("r02_exp9/c2_f32_spatial.json", "r02_bench_tile_order_position_space_maps.json"),
                     ("r02_exp11/c2_f32.json", "r02_bench_tile_order_zorder.json"),
                     ("r02_exp13/c2_f32.json", "r02_bench_tile_order_default.json"),
                     ("r02_exp13/unet_f32.json", "r02_bench_minkunet34c_f32_tile_order_default.json"),
                     ("r02_exp13/unet_f32_fresh.json", "r02_bench_minkunet34c_f32_fresh_tile_order_default.json"),
                     ("r02_exp13/unet_f32_fresh_rows.json", "r02_bench_minkunet34c_f32_fresh_row_tiles.json"),
                     ("r02_exp8/c5_split.json", "r02_bench_conv4d_split_forced.json"),
                     ("r02_exp8/c5_auto.json", "r02_bench_conv4d_split_auto.json"),
                     ("r02_exp12/rows.log", "r02_host_time_layer_step_bf16.log"),
                     ("r02_exp12/sparse_rows.log", "r02_host_time_layer_step_sparse_f32.log")):
        copy(src, dst)
    copy("r02_unet_prof_final/kernel_stats_bf16.csv", "r02_rocprof_kernel_stats_minkunet34c_bf16_final.csv")
    for extra in sys.argv[1:]:          # "src:dst" pairs for later sessions
        s, d = extra.split(":")
        copy(s, d)


if __name__ == "__main__":
    main()
