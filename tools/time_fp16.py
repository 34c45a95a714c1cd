"""Developer aid: configs[4] -- peach-forest-65, float32 vs half-precision storage: UNet stage and whole pipeline per cloud."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch, bench
from smart_tree_amd import profiling
from smart_tree_amd.data_types.cloud import Cloud
from smart_tree_amd.dataset.augmentations import AugmentationPipeline, CentreCloud
from smart_tree_amd.model.model_inference import ModelInference
from smart_tree_amd.pipeline import Pipeline
from smart_tree_amd.skeleton.skeletonize import Skeletonizer
from smart_tree_amd.synthetic import sample_tree_cloud
dev = torch.device("cuda:0")
peach = bench.WEIGHTS.parent / "peach-forest-65.npz"
npts = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
vox = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
fol = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
c = sample_tree_cloud(npts, seed=3 if fol else 0, **({"foliage_fraction": fol} if fol else {}))
cloud = Cloud(xyz=torch.from_numpy(c["xyz"]).to(dev), rgb=torch.from_numpy(c["rgb"]).to(dev))
for fp16 in (False, True):
    mi = ModelInference("unused", peach, voxel_size=vox, block_size=4, buffer_size=0.4, device=dev, fp16=fp16)
    sk = Skeletonizer(K=16, min_connection_length=0.02, minimum_graph_vertices=32, device=dev)
    pipe = Pipeline(AugmentationPipeline([CentreCloud()]), mi, sk, repair_skeletons=True, smooth_skeletons=True,
                    smooth_kernel_size=11, prune_skeletons=True, min_skeleton_radius=0.01, min_skeleton_length=0.02, device=dev)
    for _ in range(2): pipe.process_cloud(cloud=cloud)
    torch.cuda.synchronize()
    profiling.enable(True)
    t0 = time.perf_counter()
    for _ in range(6): pipe.process_cloud(cloud=cloud)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 6
    st = profiling.stage_ms(6)
    rows = profiling.kernel_table()
    profiling.enable(False)
    conv_ms = sum(v["total_ms"] for k, v in rows.items() if k.startswith("k_sparse_conv")) / 6
    print(f"fp16={fp16}: {dt*1e3:.2f} ms per cloud; unet stage {st.get('unet')} ms; conv kernels {conv_ms:.3f} ms per cloud", flush=True)
