"""bench.py -- DSPO BA-update iters/sec (+ rendered rays/sec) on the synthetic 640x480
keyframe graph G8 of BASELINE.md.

    python bench.py --gpus 1 --steps 20 --warmup 3

A "step" is one `FactorGraph.update()` equivalent on graph G8 (8 keyframes, 36 edges, 60x80):
reproject -> 4-level correlation lookup -> ConvGRU update operator -> dense BA (2 GN
iterations) -> convex upsampling.  Inputs are resident in HBM before the timed region.
With N > 1 GPUs (torch.distributed, one rank per GPU) the graph grows to 6N+2 keyframes = 36N
edges sharded by source keyframe (weak scaling, `value` in G8-sized updates per second); the
rendered frame is split N ways.  Prints ONE JSON line (see the driver contract in the task
statement).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
PMC_SUMMARY = "r06_pmc_kernels.json"   # refreshed per round by tools/pmc_passes.sh
PMC_CORR_SUMMARY = "r06_pmc_corr.json"  # tools/pmc_corr.sh
KERNEL_STATS = "r06_final_bench_kernel_stats.csv"   # rocprofv3 --kernel-trace --stats of `bench.py --g8-only` (tools/final_profile.sh)
KNN_LAYOUT = "image"                             # --knn-layout
KNN_PRODUCT_TRAFFIC = None
MFMA_F16_PEAK_TF = 2500.0  # same guide: ~2.5 PFLOP/s dense f16/bf16 (not the 2:1-sparsity figure)


def make_cfg(H=480, W=640, buffer=16, device="cuda:0"):
    return {
        "cam": {"H_out": H, "W_out": W},
        # multiview_filter.thresh: the shipped 0.01 assumes a trained update operator.  With default-init
        # weights (no checkpoint offline) the flow revisions are noise, the 1 % two-view depth check
        # rejects > 80 % of every frame and each depth_scale step would take the stage-1 fallback --
        # the bench would never run stage 2 (BA_with_scale_shift).  0.25 keeps > 80 % of the pixels.
        "tracking": {"buffer": buffer, "backend": {"BA_type": "DSPO"}, "mono_thres": 0.1,
                     "multiview_filter": {"thresh": 0.25, "visible_num": 2}, "store_images": False},
        "device": device, "setting": "bench", "scene": "G8", "data": {"output": "/tmp"},
    }


def build_graph(device, K=8, h=60, w=80, rank=0, world=1, corr_impl="volume", use_graphs=False):
    import glorie_slam_amd.synth as synth
    from glorie_slam_amd.depth_video import DepthVideo
    from glorie_slam_amd.factor_graph import FactorGraph
    from glorie_slam_amd.droid_net import UpdateModule

    g = synth.keyframe_graph(K=K, h=h, w=w, radius=3)
    fmaps, nets, inps = synth.feature_maps(K, h, w)
    video = DepthVideo(make_cfg(8 * h, 8 * w, buffer=max(K, 8), device=str(device)))
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
    video.poses[:K] = t(g["poses"][:K])
    video.disps[:K] = t(g["disps"][:K])
    video.intrinsics[:] = t(g["intrinsics"][0])
    video.fmaps[:K] = t(fmaps)
    video.nets[:K] = t(nets)
    video.inps[:K] = t(inps)
    video.counter.value = K
    # mono prior: affine-distorted true disparities + 2 % noise, 10 % holes
    rng = np.random.default_rng(synth.SEED + 9)
    sc = rng.uniform(0.5, 2.0, K).astype(np.float32)
    sq = rng.uniform(-0.05, 0.05, K).astype(np.float32)
    mono = (g["disps"][:K] - sq[:, None, None]) / sc[:, None, None] * (1 + 0.02 * rng.standard_normal((K, h, w)))
    mono[rng.uniform(size=mono.shape) < 0.1] = 0.0
    video.mono_disps[:K] = t(mono.astype(np.float32))
    torch.manual_seed(43)
    net = UpdateModule().to(device).eval()
    graph = FactorGraph(video, net, device=str(device), corr_impl=corr_impl, max_factors=-1,
                        use_graphs=use_graphs)
    sel = np.ones(len(g["ii"]), bool)
    if world > 1:   # edges sharded by source keyframe (glorie_slam_amd.dist)
        from glorie_slam_amd import dist as gdist
        owner = gdist.shard_frames(g["ii"], world)
        sel = gdist.local_edges(g["ii"], owner, rank)
        video.enable_sharding(owner, rank, world)
    graph.add_factors(t(g["ii"][sel]), t(g["jj"][sel]))
    # BA targets = reprojection + N(0, 0.5 px); weights ~ U(0,1)  
    graph.target = graph.target + t(g["noise"][sel]).permute(0, 2, 3, 1)[None]
    graph.weight = t(g["weight"][sel]).permute(0, 2, 3, 1)[None].contiguous()
    return g, video, graph


def render_cfg(device):
    return {"device": str(device),
            "pointcloud": {"nn_weighting": "distance", "use_dynamic_radius": True, "min_nn_num": 2,
                           "nn_num": 8, "radius_query": 0.08, "radius_add": 0.04, "radius_min": 0.02},
            "rendering": {"N_surface": 10, "near_end_surface": 0.95, "far_end_surface": 1.05,
                          "sample_near_pcl": True, "sigmoid_coef": 0.1, "near_end": 0.3},
            "model": {"encode_rel_pos_in_col": True, "encode_viewd": True, "c_dim": 32}}


class _Cam:
    H, W, fx, fy, cx, cy = 480, 640, 320.0, 320.0, 319.5, 239.5


def build_renderer(device, rank=0, world=1):
    """cloud PC (524,288 points) + one 640x480 view; rays are sharded over ranks by image rows"""
    import glorie_slam_amd.synth as synth
    from glorie_slam_amd.decoder import POINT
    from glorie_slam_amd.neural_point import NeuralPointCloud
    from glorie_slam_amd.renderer import Renderer
    cfg = render_cfg(device)
    pts, geo, col = synth.box_cloud()
    pts, geo, col = pts[:524288], geo[:524288], col[:524288]
    ro, rd, depth, radius, c2w = synth.box_rays()
    R = ro.shape[0]
    lo, hi = rank * R // world, (rank + 1) * R // world
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(device)
    npc = NeuralPointCloud(cfg)
    npc.add_points(t(pts), t(geo), t(col))
    torch.manual_seed(43)
    dec = POINT(cfg, use_view_direction=True).eval().to(device)
    ren = Renderer(cfg, _Cam())
    assert lo % 640 == 0 and hi % 640 == 0, "ranks take whole image rows"
    rays = dict(o=t(ro[lo:hi]), d=t(rd[lo:hi]), depth=t(depth[lo:hi]), radius=t(radius[lo:hi]), W=640)
    return npc, dec, ren, rays


def render_pass(npc, dec, ren, rays, device, two_streams=True):
    # the batching of Renderer.render_img: whole 16-row strips, so the neighbour search can walk image patches
    bs = ren.ray_batch_size
    W = rays["W"]
    image_w = W if KNN_LAYOUT == "image" else None
    if image_w:
        bs -= bs % (16 * W)
    n = rays["o"].shape[0]
    with torch.no_grad():
        ren.prepare_frame(npc, dec, device)     # shared lazily-built state on THIS stream, before the batches fork
        # the decoders' range guard is read once per frame, behind the last batch (as Renderer.render_img does for its
        # strips); a frame that tripped it is rendered again with the per-batch check and its exact-fp32 fallback
        # ... and the batches of the frame alternate between the renderer's two batch streams (Renderer.batch_stream)
        for defer in (True, False):
            for k, i in enumerate(range(0, n, bs)):
                with torch.cuda.stream(ren.batch_stream(k, device) if (two_streams and defer and n > bs) else torch.cuda.current_stream()):
                    ren.render_batch_ray(npc, dec, rays["d"][i:i + bs], rays["o"][i:i + bs], device, "color",
                                         gt_depth=rays["depth"][i:i + bs], npc_geo_feats=npc.geo_feats,
                                         npc_col_feats=npc.col_feats, cloud_pos=npc.cloud_pos(),
                                         dynamic_r_query=rays["radius"][i:i + bs], image_w=image_w, defer_guard=defer)
            ren.join_batches(device)
            if not (defer and dec.range_guard(torch.device(device)).tripped()):
                break
    return n


def _pmc_traffic():
    """per-launch HBM bytes of the profiled kernels from the committed PMC summary (separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, corrected as MI355X_MICROARCH.md prescribes;
    see profiles/README.md).  None when the summary is absent.  -> (conv, corr, knn)"""
    path = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    try:
        with open(path) as f:
            d = json.load(f)
    except Exception:
        return None, None, None
    get = lambda k: d[k]["hbm_bytes"] if k in d and d[k].get("hbm_bytes") is not None else None
    # (search, stand-alone two-table gather): the launches `roofline_knn` times live; (search, mlp_geo, mlp_nb): the product's
    # R1 + R2 launches (the decoder kernels' counters include their other traffic: positions, masks, the 32-float colour
    # feature they hand on) for the `product_gather_from_profile` side field
    knn = (get("knn_query") + get("idw_gather")) if (get("knn_query") is not None and get("idw_gather") is not None) else None
    global KNN_PRODUCT_TRAFFIC
    KNN_PRODUCT_TRAFFIC = (get("knn_query") + get("mlp_geo") + get("mlp_nb")) \
        if all(get(k) is not None for k in ("knn_query", "mlp_geo", "mlp_nb")) else None
    return get("conv_igemm_gru_zr"), get("corr_lookup"), knn


def _profile_mean_ms(substr):
    """average duration (ms) of the kernel whose name contains `substr` in the round's committed rocprofv3 kernel stats
    (profiles/<round>_final_bench_kernel_stats.csv: the whole profiled `bench.py --g8-only` run), or None"""
    import csv
    try:
        with open(os.path.join(ROOT, "profiles", KERNEL_STATS)) as f:
            for row in csv.DictReader(f):
                if substr in row["Name"]:
                    return float(row["AverageNs"]) * 1e-6, int(row["Calls"])
    except Exception:
        pass
    return None, None


def gru_gate_conv_workload(device, N, ht, wd, ii=None):
    """the dominant kernel of a BA-update step in isolation: the merged convz|convr 3x3 convolution
    of the ConvGRU as the step launches it - 320 -> 256 channels ([net | corr | flow]; the 128 context
    channels are folded into the per-pixel `pre` term once per edge set) with the gate
    epilogue - on the step's shapes; returns a launcher and its executed FLOPs
    (2 * pixels * 9 * 320 * 256; SURVEY.md 8(d): the update operator is MFMA work)"""
    from glorie_slam_amd import update_ops as U
    gen = torch.Generator(device="cpu").manual_seed(5)
    cl = lambda c: torch.randn(N, c, ht, wd, generator=gen).to(device).half().contiguous(memory_format=torch.channels_last)
    net, hx, pre = cl(128), cl(320), cl(384)
    wzr = U.pack_conv_igemm((torch.randn(256, 320, 3, 3, generator=gen) / 53.0).to(device))
    terms = torch.randn(N, 384, generator=gen).to(device)
    z, rnet = torch.empty_like(net), torch.empty_like(net)
    pre_map = None
    if ii is not None:
        # as the step launches it: one map of the context term per source keyframe, shared by its edges
        frames, ix = torch.unique(ii, sorted=True, return_inverse=True)
        pre, pre_map = pre[:frames.shape[0]].contiguous(memory_format=torch.channels_last), ix.to(torch.int32).contiguous()

    def launch():
        U.conv_igemm(net, hx[:, 128:320], wzr, 9, 256, z, epilogue=U.EPI_GRU_ZR, terms=terms[:, :256], net=net,
                     out2=rnet, pre=pre[:, 0:256], pre_map=pre_map)

    return launch, 2.0 * N * ht * wd * 9 * 320 * 256


def _cpu_lookup_edge(args):
    """worker of the CPU baseline's lookup leg: the oracle's 4-level lookup of ONE edge (numpy)"""
    from oracle import corr as ocorr
    levels, coords_n = args
    ocorr.corr_lookup_pyramid(levels, coords_n, 3)
    return 0


def cpu_baseline_rays(n_rays=5120, workers=-1):
    """exact 8-NN of the ray samples with a k-d tree over the 524k-point cloud (scipy cKDTree, all cores; the oracle's
    brute-force search is the parity checker, not a fair baseline) + torch-CPU decoders (all threads) + oracle compositing
    on `n_rays` rays spread over the frame -> (rays/s, seconds, seconds of the tree build, not charged)"""
    import glorie_slam_amd.synth as synth
    from glorie_slam_amd.decoder import POINT
    from oracle import knn as oknn
    pts, geo, col = synth.box_cloud()
    pts, geo, col = pts[:524288], geo[:524288], col[:524288]
    ro, rd, depth, radius, _ = synth.box_rays()
    sel = np.linspace(0, ro.shape[0] - 1, n_rays).astype(np.int64)
    S = 10
    z = depth[sel, None] * np.linspace(0.95, 1.05, S, dtype=np.float32)[None]
    p = (ro[sel, None] + rd[sel, None] * z[..., None]).reshape(-1, 3).astype(np.float32)
    rq = np.repeat(radius[sel], S)[:, None]
    cfg = render_cfg("cpu")
    torch.manual_seed(43)
    dec = POINT(cfg, use_view_direction=True).eval()
    cloud_t = torch.from_numpy(pts)
    t_b = time.perf_counter()
    tree = oknn.build_kdtree(pts)                # per map update, not per frame: reported, not charged
    t_build = time.perf_counter() - t_b

    class NPC:
        def get_radius_query(self):
            return 0.08

        def cloud_pos(self):
            return cloud_t

        def find_neighbors_faiss(self, pos, step='query', dynamic_radius=None, **kw):
            D, I = oknn.knn_kdtree(tree, pos.numpy(), 8, workers=workers)
            D, I = torch.from_numpy(D), torch.from_numpy(I)
            return D, I, (D < dynamic_radius.reshape(-1, 1) ** 2).sum(-1).int()

    t0 = time.perf_counter()
    with torch.no_grad():
        raw, *_ = dec(torch.from_numpy(p)[None], NPC(), "color", torch.from_numpy(geo), torch.from_numpy(col),
                      pts_num=S, cloud_pos=cloud_t, pts_views_d=torch.from_numpy(np.repeat(rd[sel], S, 0)),
                      dynamic_r_query=torch.from_numpy(rq))
        oknn.composite(raw.reshape(n_rays, S, 4).numpy(), z)
    dt = time.perf_counter() - t0
    return n_rays / dt, dt, t_build


def cpu_baseline_step(g=None):
    """Oracle ("port") timing of ONE whole BA-update step on the full graph G8 (36 edges, 60x80), un-scaled, on ALL host cores
    (SURVEY 8(d)): oracle reproject + oracle 4-level correlation lookup (one edge per worker process) + the update operator
    as the plain fp32 torch-CPU module (all threads) + oracle BA (2 GN iterations, numpy vectorised over the pixels) +
    oracle convex upsampling; then the ray leg (cpu_baseline_rays).  Runs in a process that never touches the GPU
    (bench.py --cpu-baseline-only), so the worker pool can fork."""
    import multiprocessing as mp
    import glorie_slam_amd.synth as synth
    from oracle import corr as ocorr, ba as oba, geom as ogeom
    from glorie_slam_amd.droid_net import UpdateModule
    if g is None:
        g = synth.keyframe_graph(K=8, h=60, w=80, radius=3)
    h, w, N, K = g["h"], g["w"], len(g["ii"]), g["K"]
    cores = os.cpu_count() or 1
    nproc = max(1, min(cores, N))
    rng = np.random.default_rng(0)
    t_all = time.perf_counter()
    # one edge's pyramid stands for all 36 (61 MB each; the values do not change the cost of the lookup); the workers inherit
    # it through fork (no pickling of the volume)
    levels = [rng.standard_normal((1, h, w, h >> l, w >> l)).astype(np.float16) for l in range(4)]
    torch.manual_seed(43)
    net = UpdateModule().eval()
    x = lambda c: torch.randn(1, N, c, h, w)
    xin = (x(128), x(128), x(196), x(4))
    up = (rng.standard_normal((K, 576, h, w))).astype(np.float16)
    global _CPU_LEVELS
    _CPU_LEVELS = levels
    pool = mp.get_context("fork").Pool(nproc) if nproc > 1 else None
    pmap = pool.map if pool is not None else (lambda f, it: list(map(f, it)))
    pmap(_cpu_noop, range(nproc))                # workers started and imported before anything is timed
    t0 = time.perf_counter()
    coords, _ = ogeom.reproject(g["poses"], g["disps"], g["intrinsics"], g["ii"], g["jj"])
    tgt = (coords.transpose(0, 3, 1, 2) + g["noise"]).astype(np.float32)
    pmap(_cpu_lookup_edge_inherited, [np.ascontiguousarray(coords[n:n + 1].transpose(0, 3, 1, 2)) for n in range(N)])
    t_corr = time.perf_counter() - t0
    with torch.no_grad():
        ii = torch.from_numpy(g["ii"])
        net(*xin, ii, ii)                        # (first call: oneDNN primitive creation, not the steady state)
        t1 = time.perf_counter()
        net(*xin, ii, ii)
        t_upd = time.perf_counter() - t1
    t1 = time.perf_counter()
    # (the BA's per-edge terms were tried on the worker pool as well: shipping the [6, HW] blocks back costs 8x what the
    # vectorised numpy evaluation takes - 2.0 s against 0.24 s - so this leg stays in one process)
    oba.ba(g["poses"], g["disps"], g["intrinsics"][0], tgt, g["weight"], g["eta"], g["ii"], g["jj"], 1, K, 2, 1e-4, 0.1)
    ogeom.cvx_upsample(g["disps"][:K], up, np.float16)
    t_ba = time.perf_counter() - t1
    if pool is not None:
        pool.close()
        pool.join()
    step_s = t_corr + t_upd + t_ba
    n_rays = 5120
    rays_s, rays_dt, tree_s = cpu_baseline_rays(n_rays)
    return dict(value=1.0 / step_s, unit="BA-update iters/s", cores=cores, kind="port",
                threads={"lookup_processes": nproc, "update_operator_torch_threads": torch.get_num_threads(),
                         "ba_processes": 1, "knn_kdtree_workers": cores,
                         "decoder_torch_threads": torch.get_num_threads()},
                legs_s={"reproject_lookup": t_corr, "update_operator": t_upd, "ba_2gn_upsampling": t_ba, "rays": rays_dt,
                        "kdtree_build_not_charged": tree_s},
                rays_per_sec=rays_s,
                sample=f"ONE un-scaled step on G8 ({N} edges, {h}x{w}) on {cores} host cores: oracle reproject + 4-level lookup "
                       f"{t_corr:.2f}s (numpy, one edge per process, {nproc} processes), fp32 torch-CPU update operator "
                       f"{t_upd:.2f}s ({torch.get_num_threads()} threads, second call), oracle BA 2 GN iterations + upsampling "
                       f"{t_ba:.2f}s (numpy vectorised over the pixels, 1 process); rays: {n_rays} rays spread "
                       f"over the frame, exact 8-NN by scipy cKDTree over the 524k-point cloud (workers=-1) + torch-CPU "
                       f"decoders + compositing ({rays_dt:.1f}s; tree build {tree_s:.1f}s not charged); "
                       f"{time.perf_counter() - t_all:.1f}s of CPU work in total")


_CPU_LEVELS = None


def _cpu_noop(i):
    import oracle.corr, oracle.ba  # noqa: F401,E401
    return i


def _cpu_lookup_edge_inherited(coords_n):
    return _cpu_lookup_edge((_CPU_LEVELS, coords_n))


def cpu_baseline_subprocess(timeout=600):
    """run the CPU baseline in a fresh interpreter (no HIP runtime in the process that forks the worker pool)"""
    import subprocess
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT, env=dict(os.environ, HIP_VISIBLE_DEVICES=""))
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": (r.stderr or r.stdout)[-400:]}
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="print the cpu_baseline object and exit (no GPU work)")
    ap.add_argument("--no-sequence", action="store_true", help="skip the 13-frame tracking + mapping sequence (config 3)")
    ap.add_argument("--no-strong", action="store_true", help="skip the fixed 128-keyframe graph (strong-scaling figure)")
    ap.add_argument("--soak", type=int, default=300, help="untimed steps between the burst figure and the timed steps")
    ap.add_argument("--sustained-steps", type=int, default=400, help="steps of each of the two sustained figures (tests shorten it)")
    ap.add_argument("--strong-k", type=int, default=128, help="keyframes of the strong-scaling graph (512 = long end of config 4)")
    ap.add_argument("--corr-impl", default="volume", choices=["volume", "otf"], help="correlation operator of the timed graph")
    ap.add_argument("--knn-layout", default="image", choices=["image", "linear"], help="query order of the renderer's search")
    ap.add_argument("--g8-only", action="store_true",
                    help="only the G8 / 640x480 workloads (no 40x80 graph, no frontend configuration, no sequences, no strong-scaling "
                         "graph): what tools/final_profile.sh profiles, so that a kernel's average in the stats is a G8 figure")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_step()))
        return
    global KNN_LAYOUT
    KNN_LAYOUT = args.knn_layout
    if args.g8_only:
        args.no_sequence = args.no_strong = True

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)   # (testing only: several ranks may share one GPU with gloo)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        backend = os.environ.get("GLORIE_DIST_BACKEND", "nccl")   # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    # Workload.  1 GPU: graph G8 of BASELINE.md (8 keyframes, 36 edges) -- the configuration the metric is
    # quoted on.  N GPUs: the same sliding-window topology over 6N+2 keyframes = 36N edges, sharded by source
    # keyframe (36 edges per GPU on average: per-GPU work fixed -> "weak" scaling); one step then is one
    # BA-update of the whole graph = N G8-sized updates, and `value` counts G8-sized updates per second.
    # A 1.5 ms step of a 36-edge graph is below the launch + collective floor of any multi-GPU split, the
    # sharding exists for the long graphs of BASELINE configs 4/5.
    K_graph = 6 * world + 2
    # the launches of a step are replayed as a hipGraph per (edge set, stage); when sharded the replay stops
    # before the BA, whose all-reduce and row exchange are issued eagerly (FactorGraph.update)
    g, video, graph = build_graph(device, K=K_graph, rank=rank, world=world,
                                  corr_impl=args.corr_impl,
                                  use_graphs=os.environ.get("GLORIE_NO_GRAPHS") is None)
    K = g["K"]
    poses0, disps0 = video.poses.clone(), video.disps.clone()
    step_no = [0]
    target0, weight0, net0 = graph.target.clone(), graph.weight.clone(), graph.net.clone()

    def reset():
        video.poses.copy_(poses0)
        video.disps.copy_(disps0)
        step_no[0] = 0
        graph.target, graph.weight, graph.net = target0.clone(), weight0.clone(), net0.clone()

    def make_step(graph_, video_, K_, counter):
        def step_():
            # DSPO schedule of the frontend (frontend.py:50-53): stages alternate
            opt = "pose_depth" if counter[0] % 2 == 0 else "depth_scale"
            fu = graph_.fast_update
            if counter[0] % 12 == 0 and fu is not None and (fu._pre is not None or fu._pre_kf is not None):
                # The gate convolutions over the context features are evaluated once per edge set, not per
                # iteration (FusedUpdate.precompute_shared_context: one map per source keyframe).  The frontend adds
                # one keyframe and a few edges every 12 iterations (frontend.py:23-24); the bench graph never changes,
                # so that cost is charged here explicitly, conservatively for ALL keyframes, every 12 timed steps.
                if fu._pre_kf is not None:
                    fu.precompute_shared_context((video_.inps, graph_._unique_ii(), graph_._groups()[0]))
                else:
                    fu.precompute_context()
            counter[0] += 1
            graph_.update(t0=1, t1=K_, itrs=2, use_inactive=False, opt_type=opt)
        return step_

    step = make_step(graph, video, K, step_no)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # setup, untimed: each stage is seen once eagerly and once for capture before it replays
    for _ in range(4 if graph.use_graphs else 0):
        step()
    reset()
    for _ in range(args.warmup):
        step()
    reset()

    def timed_steps(n):
        barrier()
        t0_ = time.perf_counter()
        for _ in range(n):
            step()
        barrier()
        dt = time.perf_counter() - t0_
        if world > 1:
            tmax = torch.tensor([dt], device=device)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    # `value` is measured on a WARM chip: K steps straight after the warm-up run on boost clocks (a 20-step burst is 20 ms),
    # which no tracking session sustains.  The burst figure is kept as `burst_value`; then `soak` untimed steps bring the
    # part to the clocks it holds under this load and the K timed steps of the contract follow.
    fb_burst0 = int(video.stage2_fallbacks)
    burst_elapsed = timed_steps(args.steps)
    fallbacks_burst = int(video.stage2_fallbacks) - fb_burst0
    soak_steps = int(args.soak)
    for _ in range(soak_steps):
        step()
    # The soak is there for the chip's clocks, not for the state: the update operator has default-init weights (no checkpoint
    # offline), so a trajectory that is iterated for hundreds of steps on ONE fixed graph drifts until every depth_scale
    # stage fails its mono_thres test and takes the stage-1 fallback (round 4: all 10 timed depth_scale steps did) - the
    # tracker never does that, it gets a new keyframe every 12 iterations (frontend.py:23-24).  So the state is put back
    # (untimed) and the timed steps run the metric's workload - alternating pose_depth / depth_scale with
    # BA_with_scale_shift solving (depth_video.py:268-294) - on the warm chip; fallbacks are counted INSIDE the timed region.
    reset()
    fb0 = int(video.stage2_fallbacks)
    elapsed = timed_steps(args.steps)
    fallbacks_timed = int(video.stage2_fallbacks) - fb0
    # ---- the timed steps did the work they claim: solver status, finite state, stage-2 fallbacks ----
    ba_st = video.ctx().ba_status()
    assert ba_st[0] == 0, f"BA status word after the timed loop: {ba_st} (bit 0 eta/M mismatch, bit 2 Cholesky failed)"
    assert bool(torch.isfinite(video.poses).all()) and bool(torch.isfinite(video.disps[:K]).all()), "non-finite state"
    assert bool(torch.isfinite(graph.target).all()) and bool(torch.isfinite(graph.net.float()).all())
    assert not torch.equal(video.poses[1:K], poses0[1:K]), "the timed steps did not move the poses"
    stage2_steps = (args.steps // 2)
    # ---- sustained figure: >= 400 back-to-back steps (the K-step figure above runs on boost clocks), twice:
    # (a) with the state put back every `reset_every` steps INSIDE the timed region (five device copies, 46 MB) so that
    #     stage 2 keeps solving - the metric's workload, sustained; (b) free-running, where the drifted state makes most
    #     depth_scale steps fall back (prep + host poll + eager stage-1 BA: a fallback step costs MORE than a stage-2 step)
    sustained_steps = max(int(args.sustained_steps), args.steps)
    reset_every = 48

    def sustained(periodic_reset):
        reset()
        fbs = int(video.stage2_fallbacks)
        barrier()
        t_s = time.perf_counter()
        for i_ in range(sustained_steps):
            if periodic_reset and i_ and i_ % reset_every == 0:
                reset()
            step()
        barrier()
        ms_ = 1e3 * (time.perf_counter() - t_s) / sustained_steps
        if world > 1:
            tm = torch.tensor([ms_], device=device)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            ms_ = float(tm.item())
        st_ = video.ctx().ba_status()
        assert st_[0] == 0 and bool(torch.isfinite(video.poses).all()), \
            f"after {sustained_steps} sustained steps: BA status {st_}, finite poses {bool(torch.isfinite(video.poses).all())}"
        return ms_, int(video.stage2_fallbacks) - fbs

    sustained_ms, fallbacks_sustained = sustained(True)
    sustained_free_ms, fallbacks_sustained_free = sustained(False)

    # ---- the same graph at the size of the SHIPPED Replica config (configs/Replica/replica.yaml:55-56: 320 x 640 output ->
    # 40 x 80 maps, HW = 3200; SURVEY 8: "report both"), same schedule, soak + 100 timed steps
    replica_yaml = None
    if world == 1 and not args.g8_only:
        gR, videoR, graphR = build_graph(device, K=8, h=40, w=80, use_graphs=os.environ.get("GLORIE_NO_GRAPHS") is None)
        stepR = make_step(graphR, videoR, 8, [0])
        for _ in range(4 + 150):
            stepR()
        torch.cuda.synchronize()
        t_r0 = time.perf_counter()
        for _ in range(100):
            stepR()
        torch.cuda.synchronize()
        ms_r = 1e3 * (time.perf_counter() - t_r0) / 100
        okR = videoR.ctx().ba_status()[0] == 0 and bool(torch.isfinite(videoR.poses).all())
        replica_yaml = {"workload": "G8 topology (8 keyframes, 36 edges) at 40x80 maps = 320x640 / 8", "hw": 3200, "steps": 100,
                        "ms_per_step": ms_r, "iters_per_sec": 1e3 / ms_r, "state_ok": bool(okR)}
        del graphR, videoR, gR, stepR
        torch.cuda.empty_cache()

    # ---- the exchange step by itself (self-verification of a multi-GPU record): who took part, how many bytes, how long
    exchange = None
    try:
        from glorie_slam_amd import dist as gdist
        P_ = K - 1
        n6 = 6 * P_
        hv_ = torch.zeros(n6 * n6 + n6, dtype=torch.float64, device=device)
        ctx_ = video.ctx()
        native_world = gdist.ctx_comm_world(ctx_)
        grp = video.shard["group"] if getattr(video, "shard", None) else None
        packed = n6 >= gdist.PACK_MIN_N6
        nbytes = 8 * ((n6 * (n6 + 1) // 2 + n6) if packed else (n6 * n6 + n6))
        ar_ms = None
        if world > 1 or native_world > 0:
            for _ in range(3):
                gdist.allreduce_system(hv_, grp, n6=n6, force=True, ctx=ctx_)
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            barrier()
            a0.record()
            for _ in range(20):
                gdist.allreduce_system(hv_, grp, n6=n6, force=True, ctx=ctx_)
            a1.record()
            torch.cuda.synchronize()
            ar_ms = a0.elapsed_time(a1) / 20
        edges_by_rank = [int(graph.ii.numel())]
        if world > 1:
            tl = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
            dist.all_gather(tl, torch.tensor([int(graph.ii.numel())], dtype=torch.int64, device=device))
            edges_by_rank = [int(t_.item()) for t_ in tl]
        exchange = {"torch_distributed_world": int(dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else 1,
                    "backend": (dist.get_backend() if (dist.is_available() and dist.is_initialized()) else None),
                    "rccl_ranks_ctx_communicator": int(native_world),
                    "path": "glorie_allreduce_normal_eq (ctx-owned RCCL communicator)" if native_world > 0 else
                            ("torch.distributed all_reduce" if world > 1 else "none (one rank)"),
                    "pose_unknowns": int(n6), "allreduce_bytes": int(nbytes), "packed_lower_triangle": bool(packed),
                    "allreduce_ms": ar_ms, "allreduces_per_step": 2, "edges_local_by_rank": edges_by_rank}
    except Exception as exc:
        exchange = {"error": repr(exc)[:300]}

    # ---- strong scaling: ONE fixed long graph whatever the number of ranks (the case the sharding is for, BASELINE
    # config 4: 30x40 maps, here 128 keyframes (GLORIE_STRONG_K) in a +-3 window = 756 edges, volume-free correlation), edges sharded by
    # source keyframe, one step = one BA-update of the whole graph.  Reported next to the weak figure above.
    strong = None
    if not args.no_strong:
        try:
            KS = int(args.strong_k)
            gS, videoS, graphS = build_graph(device, K=KS, h=30, w=40, rank=rank, world=world, corr_impl="otf",
                                             use_graphs=os.environ.get("GLORIE_NO_GRAPHS") is None)

            def step_s(i):
                graphS.update(t0=1, t1=KS, itrs=2, use_inactive=False, opt_type="pose_depth" if i % 2 == 0 else "depth_scale")
            for i in range(6):
                step_s(i)
            barrier()
            t_x = time.perf_counter()
            n_s = 20
            for i in range(n_s):
                step_s(i)
            barrier()
            ms_s = 1e3 * (time.perf_counter() - t_x) / n_s
            if world > 1:
                tm = torch.tensor([ms_s], device=device)
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                ms_s = float(tm.item())
            ok_s = videoS.ctx().ba_status()[0] == 0 and bool(torch.isfinite(videoS.poses).all())
            strong = {"scaling": "strong", "keyframes": KS, "edges_total": int(len(gS["ii"])), "edges_local": int(graphS.ii.numel()),
                      "hw": 1200, "steps": n_s, "ms_per_step": ms_s, "updates_per_sec": 1e3 / ms_s, "state_ok": bool(ok_s)}
            del graphS, videoS, gS
            torch.cuda.empty_cache()
        except Exception as exc:                       # never takes the headline down
            strong = {"error": repr(exc)[:300]}

    # ---- sub-metric: Gauss-Newton iterations/s of the BA step alone (B2-B7) on G8 ----
    ba_gn_per_s = None
    frontend_cfg = None
    if world == 1:
        from glorie_slam_amd import droid_backends as db
        reset()
        step()                                   # state of a running graph: targets / weights / damping of an update
        tg, wg, dm, bi, bj = graph._ba_args[:5]
        tg, wg = tg.reshape(-1, graph.ht, graph.wd, 2).contiguous(), wg.reshape(-1, graph.ht, graph.wd, 2).contiguous()
        intr0 = video.intrinsics[0].contiguous()
        pz, dz_ = video.poses.clone(), video.disps.clone()

        def ba_only():
            pz.copy_(video.poses)
            dz_.copy_(video.disps)
            db.ba(pz, dz_, intr0, None, tg, wg, dm.reshape(-1, graph.ht, graph.wd), bi, bj, 1, K, 2, 1e-4, 0.1,
                  False, False, ctx=video.ctx(), want_updates=False, targets_hwc=True)
        for _ in range(5):
            ba_only()
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0.record()
        for _ in range(100):
            ba_only()
        b1.record()
        torch.cuda.synchronize()
        ba_ms = b0.elapsed_time(b1) / 100
        ba_gn_per_s = 2.0 / (ba_ms * 1e-3)
        assert video.ctx().ba_status()[0] == 0

        # ---- the configuration the tracking driver calls (frontend.py:50-53): use_inactive=True, t0 = t1 = None,
        # on a graph whose oldest edges have retired to the inactive set; eager launches and hipGraph replay
        fe = {}
        for tag, ug in (() if args.g8_only else (("eager", False), ("replay", True))):
            _, v2, g2 = build_graph(device, K=K_graph, use_graphs=ug)
            g2.rm_factors((g2.ii == 0) | (g2.jj == 0), store=True)
            for i in range(6):
                g2.update(None, None, use_inactive=True, opt_type="pose_depth" if i % 2 == 0 else "depth_scale")
            torch.cuda.synchronize()
            t_f = time.perf_counter()
            for i in range(40):
                g2.update(None, None, use_inactive=True, opt_type="pose_depth" if i % 2 == 0 else "depth_scale")
            torch.cuda.synchronize()
            fe[tag + "_its_per_s"] = 40 / (time.perf_counter() - t_f)
            fe["edges_active"], fe["edges_inactive"] = int(g2.ii.numel()), int(g2.ii_inac.numel())
            assert v2.ctx().ba_status()[0] == 0 and bool(torch.isfinite(v2.poses).all())
            del g2, v2
        frontend_cfg = fe or None
        torch.cuda.empty_cache()

    # ---- roofline of the dominant kernel of a step (GRU gate convolution, 23 % of it): HIP events on
    # the launch stream around back-to-back launches
    reset()
    conv_launch, conv_flops = gru_gate_conv_workload(device, graph.ii.shape[0], graph.ht, graph.wd,
                                                     graph.ii if graph.share_context else None)
    for _ in range(3):
        conv_launch()
    cv0, cv1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cv0.record()
    for _ in range(20):
        conv_launch()
    cv1.record()
    torch.cuda.synchronize()
    conv_b2b_ms = cv0.elapsed_time(cv1) / 20
    # the same launch inside real steps: events on the launch stream around the gate convolution of 30 eager BA-update
    # iterations (between the correlation encoder and the q gate, as a step runs it; this is the duration
    # `rocprofv3 --kernel-trace --stats` averages over the steps).  Twenty back-to-back launches of one MFMA-bound kernel
    # run at lower clocks and with every workgroup's prologue in phase: kept as `back_to_back_ms`.
    # (a local G8 graph on every rank: 36 edges, no collective)
    _, v3, g3 = build_graph(device, K=8, use_graphs=False)
    for i in range(4):
        g3.update(t0=1, t1=8, itrs=2, use_inactive=False, opt_type="pose_depth" if i % 2 == 0 else "depth_scale")
    g3.fast_update.gate_events = []
    g3.fast_update.q_events, g3.fast_update.heads_events = [], []
    g3.fast_update.corr_events = []
    for i in range(30):
        g3.update(t0=1, t1=8, itrs=2, use_inactive=False, opt_type="pose_depth" if i % 2 == 0 else "depth_scale")
    torch.cuda.synchronize()
    ev = g3.fast_update.gate_events
    evq, evh = g3.fast_update.q_events, g3.fast_update.heads_events
    evc = g3.fast_update.corr_events
    g3.fast_update.gate_events = g3.fast_update.q_events = g3.fast_update.heads_events = None
    g3.fast_update.corr_events = None
    corr_step_ms = (sum(a.elapsed_time(b) for a, b in evc) / len(evc)) if evc else None
    n3, hw3 = int(g3.ii.shape[0]), g3.ht * g3.wd
    q_ms = sum(a.elapsed_time(b) for a, b in evq) / max(len(evq), 1)
    heads_ms = sum(a.elapsed_time(b) for a, b in evh) / max(len(evh), 1)
    q_flops = 2.0 * n3 * hw3 * 9 * 320 * 128                      # convq over [r*net | corr | flow] (+ hoisted context term)
    heads_flops = 2.0 * n3 * hw3 * (9 * 128 * 384 + 2 * 128 * 18)  # delta[0] | weight[0] | agg.conv1 + the heads' tap GEMMs
    assert len(ev) == 30
    conv_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    conv_flops = 2.0 * int(g3.ii.shape[0]) * g3.ht * g3.wd * 9 * 320 * 256
    del g3, v3
    conv_tf = conv_flops / (conv_ms * 1e-3) / 1e12
    # ---- roofline of the correlation gather (the HBM-bound kernel north_star names)
    coords1, _ = video.reproject(graph.ii, graph.jj)
    reps = 20
    corr_traffic_key = None
    if graph.corr_impl == "otf":
        # volume-free lookup with corr_encoder[0] fused behind it, as the step launches it
        from glorie_slam_amd.droid_net import FusedLookup
        blk, rig = graph._otf_block(), graph._otf_rig
        fl = FusedLookup(blk, coords1, (rig * graph.ii).contiguous(), (rig * graph.jj).contiguous())
        fu = graph.fast_update
        c1buf = torch.empty((graph.ii.shape[0], 128, graph.ht, graph.wd), dtype=torch.float16, device=device,
                            memory_format=torch.channels_last)
        fu._sync()
        corr_fn = lambda: fl.encode_into(fu.W, c1buf)
        corr_kernel = "corr_otf8_kernel<false,true> (volume-free MFMA lookup, 8x8 source tiles, + fused corr_encoder[0])"
    elif getattr(graph.corr, "layout", None) == "dm":
        # displacement-major pyramid: lookup + corr_encoder[0] in one launch, as the step launches it
        from glorie_slam_amd.droid_net import ArenaLookup
        fu = graph.fast_update
        fu._sync()
        al = ArenaLookup(graph.corr, coords1)
        c1buf = torch.empty((graph.ii.shape[0], 128, graph.ht, graph.wd), dtype=torch.float16, device=device,
                            memory_format=torch.channels_last)
        corr_fn = lambda: al.encode_into(fu.W, c1buf)
        corr_kernel = ("corr_dm_encode_kernel<false> (fp16, displacement-major source-tiled pyramid, 4 levels, "
                       "+ corr_encoder[0] as MFMA epilogue)")
        corr_traffic_key = "dm_enc"
    else:
        corr_fn = lambda: graph.corr(coords1, channels_last=True)    # as the step launches it (FactorGraph.update)
        corr_kernel = "corr_lookup_r3_tiled_kernel (fp16, 4x8-tiled pyramid)"
    for _ in range(3):
        corr_fn()
    torch.cuda.synchronize()
    # device time of the launch: `reps` calls replayed as one hipGraph (the eager Python wrapper is slower than the kernel)
    cgraph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cgraph):
        for _ in range(reps):
            corr_fn()
    cgraph.replay()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    cgraph.replay()
    ev1.record()
    torch.cuda.synchronize()
    N = graph.ii.shape[0]
    HW = graph.ht * graph.wd
    alg_bytes = 936.0 * N * HW  # 936 B per edge-pixel
    # the python wrapper adds a coords permute/copy (small); kernel time is reported by rocprof
    # Twenty back-to-back replays of the one launch read the same 89 MB again and again (it stays in the 256 MB Infinity
    # Cache) at boost clocks: kept as `back_to_back_ms`.  `ms_per_launch` / `frac` are the launch inside real BA-update steps
    # (events around it in the 30 eager iterations above: cold pyramid, the step's clocks) - the duration
    # `rocprofv3 --kernel-trace --stats` averages over the steps.
    corr_b2b_ms = ev0.elapsed_time(ev1) / reps
    corr_ms = corr_step_ms if (corr_step_ms and graph.corr_impl != "otf" and graph.fast_update is not None
                               and getattr(graph.corr, "layout", None) == "dm") else corr_b2b_ms
    achieved = alg_bytes / (corr_ms * 1e-3) / 1e9
    corr_prof_ms, corr_prof_calls = _profile_mean_ms("corr_dm_encode_kernel" if corr_traffic_key == "dm_enc" else "corr_otf8"
                                                     if graph.corr_impl == "otf" else "corr_lookup")

    # ---- M2: rendered rays/sec (full 640x480 frame, rays sharded over ranks) ----------
    npc, dec, ren, rays = build_renderer(device, rank, world)
    render_pass(npc, dec, ren, rays, device)
    barrier()
    t_r = time.perf_counter()
    n_r = 0
    render_reps = 3
    for _ in range(render_reps):
        n_r += render_pass(npc, dec, ren, rays, device)
    barrier()
    t_r = time.perf_counter() - t_r
    if world > 1:
        tm = torch.tensor([t_r], device=device)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        t_r = float(tm.item())
    rays_total = torch.tensor([float(n_r)], device=device)
    if world > 1:
        dist.all_reduce(rays_total)
    rays_per_s = float(rays_total.item()) / t_r
    # the same frame with the decoders on the exact-fp32 MFMA kernels (what the range guard of the split kernels falls back
    # to; GLORIE_MLP_F32 is read at every launch)
    rays_per_s_f32 = None
    if world == 1:
        os.environ["GLORIE_MLP_F32"] = "1"
        try:
            render_pass(npc, dec, ren, rays, device)
            torch.cuda.synchronize()
            t_f = time.perf_counter()
            n_f = render_pass(npc, dec, ren, rays, device)
            torch.cuda.synchronize()
            rays_per_s_f32 = n_f / (time.perf_counter() - t_f)
        finally:
            del os.environ["GLORIE_MLP_F32"]
    assert not dec.range_guard(device).tripped(), "the range guard of the fp16-split decoders tripped on the bench scene"
    # M2 on the 5000-ray training batch (mapper.py:390-515 samples 5000 pixels per iteration), forward
    gsel = torch.Generator(device="cpu").manual_seed(5)
    pick = torch.randperm(rays["o"].shape[0], generator=gsel)[:5000].to(device)
    b5 = {k: v[pick].contiguous() for k, v in rays.items() if torch.is_tensor(v)}

    def batch5000():
        with torch.no_grad():
            return ren.render_batch_ray(npc, dec, b5["d"], b5["o"], device, "color", gt_depth=b5["depth"],
                                        npc_geo_feats=npc.geo_feats, npc_col_feats=npc.col_feats,
                                        cloud_pos=npc.cloud_pos(), dynamic_r_query=b5["radius"])
    for _ in range(3):
        batch5000()
    torch.cuda.synchronize()
    t_b = time.perf_counter()
    for _ in range(50):
        out5 = batch5000()
    torch.cuda.synchronize()
    batch_ms = 1e3 * (time.perf_counter() - t_b) / 50
    assert bool(torch.isfinite(out5[2]).all())
    # one mapping iteration on that batch (mapper.py:390-515): forward, L1 depth + colour loss, backward to the
    # feature tables and the decoder parameters, Adam step - on the HIP training path (csrc/train.hip) and, for
    # comparison, through torch autograd over the plain modules (what the reference's mapper runs)
    from glorie_slam_amd.render_train import FeatureAdam
    train = {}
    gt_col5 = torch.rand(5000, 3, device=device)
    for tag, hip in (("hip", True), ("torch_autograd", False)):
        geo_t = npc.geo_feats.detach().clone().requires_grad_(True)
        col_t = npc.col_feats.detach().clone().requires_grad_(True)
        dec_t = dec.train()
        opt = FeatureAdam([{"params": list(dec_t.parameters()), "lr": 1e-3}, {"params": [geo_t], "lr": 1e-3},
                           {"params": [col_t], "lr": 5e-3}])
        ren.use_train_path, dec_t.use_fused = hip, hip

        def iteration():
            opt.zero_grad()
            d5, _, c5, _, _ = ren.render_batch_ray(npc, dec_t, b5["d"], b5["o"], device, "color", gt_depth=b5["depth"],
                                                   npc_geo_feats=geo_t, npc_col_feats=col_t, cloud_pos=npc.cloud_pos(),
                                                   dynamic_r_query=b5["radius"])
            loss = torch.abs(b5["depth"] - d5).sum() + 0.5 * torch.abs(gt_col5 - c5).sum()
            loss.backward()
            opt.step()
            return loss
        for _ in range(3):
            l0 = iteration()
        torch.cuda.synchronize()
        t_t = time.perf_counter()
        n_it = 20
        for _ in range(n_it):
            l1 = iteration()
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t_t) / n_it
        assert bool(torch.isfinite(l1)) and float(l1) < float(l0), "the mapping iterations must reduce the loss"
        train[tag + "_ms_per_iteration"] = ms
        train[tag + "_rays_per_sec"] = 5000.0 / (ms * 1e-3)
    ren.use_train_path, dec.use_fused = True, True
    dec.eval()
    # ---- BASELINE config 3 in miniature: 13 frames of a synthetic 640x480 stream through the tracker + mapper composition
    # (glorie_slam_amd.pipeline: motion filter -> frontend with hipGraph replays -> periodic global BA -> 20 mapping
    # iterations per kept keyframe -> final global BA); wall-clock per stage incl. the host control flow
    sequence = None
    if rank == 0 and world == 1 and not args.no_sequence:
        from glorie_slam_amd.pipeline import synthetic_images, synthetic_runner
        Ks = 13
        srun, sc = synthetic_runner(device, Ks, zero_flow_head=True, map_iters=20, map_rays=1000, buffer=512)
        simgs = synthetic_images(Ks)
        torch.cuda.synchronize()
        t_s = time.perf_counter()
        ssum = srun.run(((k, simgs[k:k + 1]) for k in range(Ks)), sc["intrinsics"], final_ba_steps=7)
        torch.cuda.synchronize()
        t_s = time.perf_counter() - t_s
        tr = srun.timing["track_ms"]
        assert ssum["keyframes"] == Ks and ssum["mapped"] == Ks and sc["video"].ctx().ba_status()[0] == 0
        assert sum(1 for a, b in ssum["losses"] if b < a) >= Ks - 2, "mapping iterations must reduce the loss"
        sequence = {"frames": Ks, "resolution": "640x480 (60x80 BA)", "video_buffer": 512, "wall_s": t_s,
                    "bootstrap_ms": tr[7], "ms_per_kept_keyframe": float(np.median(tr[8:])),
                    "ms_per_mapping_iteration": float(np.mean(srun.timing["map_iter_ms"][2:])),
                    "ms_per_global_ba_2steps": float(np.median(srun.timing["ba_ms"])), "cloud_points": ssum["points"],
                    "note": "untrained networks, flow head zeroed (fixed point at the generating trajectory): costs, not accuracy"}
        del srun, sc, simgs
    # ---- config 3 at length: 220 frames (glorie_slam_amd.pipeline.synthetic_long_runner: culled repeats, loop closure into
    # revisits, global BA every 20 keyframes, 20 mapping iterations per kept keyframe, 512-frame buffer)
    sequence_long = None
    if rank == 0 and world == 1 and not args.no_sequence:
        try:
            from glorie_slam_amd.pipeline import synthetic_long_runner
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
            lrun, lc, lframes = synthetic_long_runner(device, n_frames=220, map_iters=20)
            n_loop = [0, 0]
            real_loop = lrun.frontend.loop_closing.loop_ba

            def counted_loop(*a, **k):
                o_ = real_loop(*a, **k)
                n_loop[0] += 1
                n_loop[1] += int(o_[1] > 0)
                return o_
            lrun.frontend.loop_closing.loop_ba = counted_loop
            torch.cuda.synchronize()
            t_l = time.perf_counter()
            lsum = lrun.run(lframes(), lc["intrinsics"], final_ba_steps=4)
            torch.cuda.synchronize()
            t_l = time.perf_counter() - t_l
            tr = np.array(lrun.timing["track_ms"])
            kp = np.array(lrun.timing["kept"])
            after_boot = np.arange(len(tr)) > 8
            kept_ms = tr[kp & after_boot]
            fgl = lrun.frontend.graph
            Kl = int(lsum["keyframes"])
            sequence_long = {
                "frames": 220, "kept_keyframes": Kl, "culled": int(220 - Kl), "resolution": "640x480 (60x80 BA)", "video_buffer": 512,
                "wall_s": t_l, "ms_per_kept_keyframe_p50": float(np.percentile(kept_ms, 50)),
                "ms_per_kept_keyframe_p95": float(np.percentile(kept_ms, 95)),
                "ms_per_culled_frame_p50": float(np.percentile(tr[~kp & after_boot], 50)) if (~kp & after_boot).any() else None,
                "loop_ba_calls": n_loop[0], "loop_ba_with_edges": n_loop[1],
                "global_ba_calls": len(lrun.timing["ba_ms"]), "ms_per_global_ba_last": float(lrun.timing["ba_ms"][-1]),
                "ms_per_mapping_iteration": float(np.mean(lrun.timing["map_iter_ms"][2:])),
                "hipgraph_captures_per_keyframe": fgl.stats["captures"] / max(Kl, 1),
                "hipgraph_replays_per_keyframe": fgl.stats["replays"] / max(Kl, 1),
                "eager_updates_per_keyframe": fgl.stats["eager"] / max(Kl, 1),
                "corr_arena_slots_high_water": int(fgl.corr.capacity), "frontend_max_factors": int(fgl.max_factors),
                "cloud_points": int(lsum["points"]),
                "max_memory_allocated_gb": torch.cuda.max_memory_allocated() / 2 ** 30,
                "state_ok": bool(lc["video"].ctx().ba_status()[0] == 0 and torch.isfinite(lc["video"].poses[:Kl]).all()),
                "mapping_loss_decreased": int(sum(1 for a_, b_ in lsum["losses"] if b_ < a_)),
                "note": "untrained networks, flow head zeroed (fixed point at the generating trajectory): costs, not accuracy"}
            del lrun, lc, lframes
            torch.cuda.empty_cache()
        except Exception as exc:
            sequence_long = {"error": repr(exc)[:300]}
    # KNN + feature gather alone (HIP events), 2156 B per sample
    S = ren.N_surface
    nq = min(rays["o"].shape[0], 61440)            # 96 image rows: the batch render_img evaluates
    z = rays["depth"][:nq, None] * torch.linspace(0.95, 1.05, S, device=device)[None]
    pq = (rays["o"][:nq, None] + rays["d"][:nq, None] * z[..., None]).reshape(-1, 3).contiguous()
    rq = rays["radius"][:nq].repeat_interleave(S)
    from glorie_slam_amd import point_ops

    img_w = int(rays["W"]) if "W" in rays else None
    layout = (S, img_w) if (img_w and args.knn_layout == "image") else None

    # R1 as the renderer launches it: ONE launch = exact search bounded by the query radius + IDW weights + neighbour mask
    # (glorie_knn_query_weights).  R2 (the 128-byte feature rows) is gathered inside the decoder kernels in the product;
    # the stand-alone two-table gather is timed next to it so that the pair can be priced against SURVEY's 2156 B per sample.
    def knn_product():
        return npc.index.search(pq, 8, radius_per_query=rq, image_layout=layout, weights=(2, False, True))

    def timed(fn, reps=5):
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    knn_search_ms = timed(knn_product)
    Dk, Ik, nnk, _, _ = knn_product()
    gather_ms = timed(lambda: point_ops.idw_gather2(Dk, Ik, nnk, npc.geo_feats, npc.col_feats, radius_per_query=rq))
    # R2 as the PRODUCT performs it: the feature rows are pulled inside mlp_geo_v4 / mlp_nb_v4, not by a stand-alone gather.
    # Their gather phases are timed LIVE in this process through the measurement instantiations of the two kernels
    # (glorie_render_mlp stage_flags & 4: ids, weights, the 8 neighbour rows of both tables, positions and the interpolation
    # stay, the networks are removed) next to the live search launch: `roofline_knn.frac` is that composite - the
    # stand-alone two-table gather (a launch the renderer never makes) is kept as a side field only
    vq_k = rays["d"][:nq].repeat_interleave(S, dim=0).contiguous()
    Dk, Ik, nnk, wk_, hask_ = knn_product()
    packed_k = dec._packed()

    def gather_phase():
        return point_ops.render_mlp(packed_k, pq, vq_k, npc.cloud_pos(), npc.col_feats, None, Ik, wk_, hask_,
                                    geo_feats=npc.geo_feats, gather_phase_only=True)
    product_gather_ms = timed(gather_phase)
    knn_bytes = 2156.0 * pq.shape[0]
    knn_ms = knn_search_ms + product_gather_ms
    knn_gbs = knn_bytes / (knn_ms * 1e-3) / 1e9
    standalone_ms = knn_search_ms + gather_ms
    # fused decoders alone: executed FLOPs (post-sum F_theta form, 358,848 FLOP per sample)
    D_, I_, nn_ = npc.index.search(pq, 8, radius_per_query=rq)
    cg_, has_, w_ = point_ops.idw_gather(D_, I_, nn_, npc.geo_feats, radius_per_query=rq, return_weights=True)
    vq = rays["d"][:nq].repeat_interleave(S, dim=0).contiguous()
    packed = dec._packed()
    point_ops.render_mlp(packed, pq, vq, npc.cloud_pos(), npc.col_feats, cg_, I_, w_, has_)
    mlp_ms = timed(lambda: point_ops.render_mlp(packed, pq, vq, npc.cloud_pos(), npc.col_feats, cg_, I_, w_, has_))
    mlp_flops = 2.0 * 179424.0 * pq.shape[0]
    mlp_tf = mlp_flops / (mlp_ms * 1e-3) / 1e12
    conv_traffic, corr_traffic, knn_traffic = _pmc_traffic()
    if corr_traffic_key is not None:
        # the correlation kernels have their own passes (tools/pmc_corr.sh: calibrated on a streaming copy AND on a
        # launch of known byte count in the kernel's own 2-byte-per-lane access pattern)
        try:
            with open(os.path.join(ROOT, "profiles", PMC_CORR_SUMMARY)) as f:
                dd = json.load(f)[corr_traffic_key]
            corr_traffic = dd["hbm_read_bytes"] + dd["hbm_write_bytes"]
        except Exception:
            corr_traffic = None
    full = world == 1 and N == 36

    out = {
        "metric": "DSPO BA-update iters/sec + rendered rays/sec, 640x480 Replica keyframe graph",
        "value": world * args.steps / elapsed,
        "unit": "BA-update iters/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "soak_steps": soak_steps, "burst_value": world * args.steps / burst_elapsed,
        "sustained_ms_per_step": sustained_ms, "sustained_steps": sustained_steps,
        "sustained_value": world * 1e3 / sustained_ms,
        "sustained_reset_every": reset_every, "sustained_stage2_fallbacks": fallbacks_sustained,
        "sustained_free_running": {"ms_per_step": sustained_free_ms, "value": world * 1e3 / sustained_free_ms,
                                   "stage2_fallbacks": fallbacks_sustained_free, "stage2_steps": sustained_steps // 2},
        "checks": {"ba_status": ba_st, "stage2_fallbacks_timed": fallbacks_timed, "stage2_steps": stage2_steps,
                   "stage2_fallbacks_burst": fallbacks_burst, "state_reset_after_soak": True, "state_finite": True},
        "ba_gn_iters_per_sec": ba_gn_per_s,
        "frontend_config": frontend_cfg,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16 corr/ConvGRU, f32 Jacobians, f64 solve; render: f32 (decoder matmuls as 3-term f16 splits, f32 accumulate)", "data": "synthetic",
        "config": {"workload": (f"G8: 8 keyframes, 36 edges, 60x80 (640x480/8), BA itrs=2, DSPO stages alternating, "
                                 f"context part of the GRU gates (one map per source keyframe) re-evaluated for all keyframes every 12 steps"
                                if world == 1 else
                                f"G8 topology over {K_graph} keyframes = {len(g['ii'])} edges (36 per GPU), 60x80, BA itrs=2, "
                                f"DSPO stages alternating; value = G8-sized (36-edge) updates per second")
                               + "; multiview_filter.thresh = 0.25 (reference ships 0.01, configs/mono_point_slam.yaml:75: with "
                                 "untrained weights 0.01 rejects > 80 % of every frame and no depth_scale step would reach stage 2)"
                               + "; render: 524k-point cloud, one 640x480 view, 10 samples/ray",
                   "edges_local": int(N), "edges_total": int(len(g["ii"])), "hw": int(HW),
                   "keyframes": int(K_graph),
                   "parallelism": ("single GPU" if world == 1 else
                                   f"edges sharded by source keyframe over {world} GPUs + RCCL all-reduce of the "
                                   f"reduced normal equations; rays of the one frame sharded {world} ways")},
        "roofline": {"bound": "mfma", "kernel": "conv_pp_kernel<EPI_GRU_ZR> (ConvGRU convz|convr over [net|corr|flow], 320->256, 3x3, + hoisted context term; 256 x 256 ping-pong tile)",
                     "achieved": conv_tf, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                     "frac": conv_tf / MFMA_F16_PEAK_TF, "traffic": conv_traffic if full else None,
                     "traffic_source": "profiles/" + PMC_SUMMARY + " (rocprofv3 --pmc passes on the builder's box, not this run)",
                     "flops_per_launch": conv_flops, "ms_per_launch": conv_ms, "back_to_back_ms": conv_b2b_ms},
        "roofline_q": {"bound": "mfma", "kernel": "conv_halo_kernel<EPI_GRU_Q,4> (ConvGRU convq, 320->128, 3x3, GRU blend epilogue; haloed 128 x 128 tile - from 262,144 pixels on conv_ppw_kernel)",
                       "achieved": q_flops / (q_ms * 1e-3) / 1e12 if q_ms else None, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s",
                       "frac": (q_flops / (q_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TF) if q_ms else None, "traffic": None,
                       "flops_per_launch": q_flops, "ms_per_launch": q_ms},
        "roofline_heads": {"bound": "mfma", "kernel": "conv_halo_kernel<EPI_HEADS,4> (delta[0] | weight[0] | agg.conv1, 128->384, "
                                                      "3x3, tap GEMMs of the heads in the epilogue)",
                           "achieved": heads_flops / (heads_ms * 1e-3) / 1e12 if heads_ms else None, "peak": MFMA_F16_PEAK_TF,
                           "unit": "TFLOP/s", "frac": (heads_flops / (heads_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TF) if heads_ms else None,
                           "traffic": None, "flops_per_launch": heads_flops, "ms_per_launch": heads_ms},
        "roofline_corr": {"bound": "hbm", "kernel": corr_kernel,
                          "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": achieved / HBM_PEAK_GBS, "traffic": corr_traffic if full else None,
                          "traffic_source": "profiles/" + (PMC_CORR_SUMMARY if corr_traffic_key else PMC_SUMMARY) + " (rocprofv3 --pmc passes on the builder's box, not this run)",
                          "alg_bytes_per_launch": alg_bytes, "ms_per_launch": corr_ms,
                          # `frac` above: the launch inside 30 live eager steps of THIS run (cold pyramid, the step's clocks).
                          # The same launch averaged by rocprofv3 over every step of the round's profiled run:
                          "profile_mean_ms": corr_prof_ms, "profile_calls": corr_prof_calls,
                          "profile_frac": (alg_bytes / (corr_prof_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (corr_prof_ms and full) else None,
                          "profile_source": "profiles/" + KERNEL_STATS,
                          # 20 replays of the one launch re-read the same 92 MB from the 256 MB Infinity Cache: NOT HBM evidence
                          "back_to_back_ms_l3_resident": corr_b2b_ms,
                          # bytes the hardware really moved (PMC) over the same time: the rocprof HBM GB/s
                          "measured_hbm_gbs": (corr_traffic / (corr_ms * 1e-3) / 1e9) if (full and corr_traffic) else None,
                          "measured_hbm_frac": (corr_traffic / (corr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
                          if (full and corr_traffic) else None},
        "rays_per_sec": rays_per_s,
        "rays_per_sec_f32_mfma": rays_per_s_f32,
        # the frame is a fixed ray budget split over the ranks (strong scaling); the BA-update graph grows with N (weak)
        "rays_scaling": "strong",
        "rays_per_sec_batch5000": 5000.0 / (batch_ms * 1e-3), "ms_per_batch5000": batch_ms,
        "train_batch5000": train,
        "sequence": sequence,
        "sequence_long": sequence_long,
        "replica_yaml_40x80": replica_yaml,
        "exchange_step": exchange,
        "strong_scaling_graph": strong,
        "render": {"rays_local": int(n_r // render_reps), "samples_per_ray": int(S), "cloud_points": int(npc.pts_num()),
                   "ms_per_frame_shard": 1e3 * t_r / render_reps},
        "roofline_knn": {"bound": "hbm", "kernel": "knn_query_kernel<8> as the renderer launches it (image-patch order, bounded by the "
                                                   "query radius, + IDW weights and mask) + the feature-pull phases of mlp_geo_v4 / "
                                                   "mlp_nb_v4 (where the product gathers the 8 neighbour rows of both tables), all "
                                                   "three timed live in this run (glorie_render_mlp stage_flags & 4)",
                         "search_ms": knn_search_ms, "product_gather_ms": product_gather_ms,
                         "achieved": knn_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": knn_gbs / HBM_PEAK_GBS,
                         "traffic": KNN_PRODUCT_TRAFFIC if world == 1 else None,
                         "traffic_source": "profiles/" + PMC_SUMMARY + " (search + the two decoder kernels incl. their other "
                                           "traffic; rocprofv3 --pmc passes on the builder's box, not this run)",
                         "alg_bytes_per_launch": knn_bytes, "ms_per_launch": knn_ms,
                         # a launch the product does NOT make (search + stand-alone two-table gather), for comparison only
                         "standalone_gather2": {"gather_ms": gather_ms, "ms": standalone_ms,
                                                "frac": knn_bytes / (standalone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                "traffic": knn_traffic if world == 1 else None}},
        # decoders: fp32-accurate matmuls as hi*hi + hi*lo + lo*hi on the fp16 matrix cores (per-neighbour and colour
        # all three kernels; the narrow output layers stay fp32).  `achieved` counts ALGORITHMIC (fp32) FLOPs; the
        # ceiling of a 3-product split is the dense fp16 peak / 3; the fp32 MFMA path it replaced peaks at 157.3.
        "roofline_mlp": {"bound": "mfma", "kernel": "mlp_geo_v4 + mlp_nb_v4 + mlp_col_v4 (fp16 MFMA 16x16x32, 3-term hi/lo split, "
                                                    "fp32 accumulate; transposed form)",
                         "achieved": mlp_tf, "peak": 2500.0 / 3.0, "unit": "TFLOP/s", "frac": mlp_tf / (2500.0 / 3.0),
                         "vs_fp32_mfma_peak": mlp_tf / 157.3,
                         "traffic": None, "flops_per_launch": mlp_flops, "ms_per_launch": mlp_ms},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_subprocess()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
