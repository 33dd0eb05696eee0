#!/usr/bin/env python3
"""Drop-in command line for the reference's depth_enhanced_reconstruction.py (its __main__ block, :1418-1467).

    python depth_enhanced_reconstruction.py --input ./input_folder/buddha_images --output ./output --fx ... 

--output is a DIRECTORY; the result is <output>/reconstruction.ply (DER:1247).  The reference computes depth with
Depth-Anything fetched by model name (DER:114-118).  That network stays upstream PyTorch-ROCm (it is not part of this back
end); two ways to get depth here:
  --depth-model DIR   a LOCAL checkpoint directory: the upstream transformers classes run it on the GPU (same call sequence
                      as DER:139-165) and the depth tensors go straight into the frame slots, no host copy;
  (default)           depth files written by the reference's depth_processor.py (`<stem>_depth.npy|png`), looked for in
                      --depth-folder, else <input>/depth, <input>_depth, <input>/depth_images, <input>.
--per-frame-ply additionally writes <output>/pointclouds/<stem>.ply per frame, as depth_processor.py does (DP:923-934).
DER's dense defaults are used: depth limits 0.1 / 100 m, subsample 4, voxel 5 mm, no outlier filter.
"""
import argparse
import os
import sys
from pathlib import Path

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    parser = argparse.ArgumentParser(description="Depth-Enhanced 3D Reconstruction")
    parser.add_argument("--input", type=str, default="./input_folder/buddha_images", help="Input folder with images")
    parser.add_argument("--output", type=str, default="./output", help="Output directory")
    parser.add_argument("--fx", type=float, default=1719.0, help="Focal length X")
    parser.add_argument("--fy", type=float, default=1719.0, help="Focal length Y")
    parser.add_argument("--cx", type=float, default=540.0, help="Principal point X")
    parser.add_argument("--cy", type=float, default=960.0, help="Principal point Y")
    parser.add_argument("--no-depth", action="store_true", help="Disable depth estimation")
    parser.add_argument("--no-hybrid", action="store_true", help="Disable hybrid features")
    # additive
    parser.add_argument("--depth-folder", type=str, default=None)
    parser.add_argument("--depth-model", type=str, default=None,
                        help="local Depth-Anything checkpoint directory (save_pretrained layout); never downloaded")
    parser.add_argument("--depth-scale", type=float, default=1.0, help="metric scale of the (relative) depth")
    parser.add_argument("--per-frame-ply", action="store_true", help="also write <output>/pointclouds/<stem>.ply per frame (DP:923-934)")
    parser.add_argument("--grid", type=int, default=1024, help="fusion volume budget: at most GRID^3 voxels in total")
    parser.add_argument("--device", type=int, default=0)
    args = parser.parse_args(argv)

    from tl3d import fileio
    from tl3d.config import ReconstructionConfig
    from tl3d.pipeline import DepthToReconstructionPipeline

    inp = Path(args.input)
    n_images = len([f for f in inp.iterdir() if f.suffix.lower() in fileio.IMAGE_SUFFIXES]) if inp.is_dir() else 0
    if n_images < 2:
        print("Need at least 2 images for reconstruction")
        return 1
    if args.no_depth:
        print("--no-depth: the sparse-only path of the reference (SIFT/ORB/LSD structure from motion) is out of scope "
              "of the device back end")
        return 1
    config = ReconstructionConfig(fx=args.fx, fy=args.fy, cx=args.cx, cy=args.cy, min_depth=0.1, max_depth=100.0,
                                  voxel_size=0.005, subsample_factor=4, grid_dim=args.grid, device=args.device,
                                  depth_scale=args.depth_scale,
                                  outlier_filter=False)                   # DER's merge_pointclouds has no outlier filter (DER:615-645)
    pipeline = DepthToReconstructionPipeline(config)
    if args.depth_model:
        from tl3d.depthnet import LocalDepthEstimator
        try:
            net = LocalDepthEstimator(args.depth_model, device=args.device)
        except FileNotFoundError as e:
            print(e)
            return 1
        files = sorted(f for f in inp.iterdir() if f.suffix.lower() in fileio.IMAGE_SUFFIXES)
        print(f"Found {len(files)} images")
        images, names = [], []
        for f in files:
            img = fileio.read_image_bgr(f)
            if img is not None:
                images.append(img)
                names.append(f.name)
        print("\n--- Step 1: Estimating depth maps ---")                  # DER:1081
        depths = []
        for i, img in enumerate(images):
            depths.append(net.estimate(img))                              # float32 [H, W] on the GPU
            print(f"  Depth {i + 1}/{len(images)}")
        pipeline.set_frames(images, depths, names)
        loaded = len(images)
        del net
    else:
        cands = [args.depth_folder] if args.depth_folder else [inp / "depth", Path(str(inp) + "_depth"), inp / "depth_images", inp]
        loaded = 0
        for c in cands:
            if c is not None and Path(c).is_dir():
                # per-frame clouds are exported from host arrays; otherwise the frames stream from the files to the GPU
                loaded = (pipeline.load_data if args.per_frame_ply else pipeline.load_data_streaming)(str(inp), str(c))
                if loaded >= 2:
                    break
        if loaded < 2:
            print("No depth maps found next to the images. Pass --depth-model <local Depth-Anything checkpoint directory> to "
                  "estimate them here (upstream PyTorch-ROCm), or run the reference's depth_processor.py (or any producer of "
                  "<stem>_depth.npy / .png) first and pass --depth-folder.")
            return 1
    if loaded < 2:
        print("Need at least 2 images for reconstruction")
        return 1
    out_dir = Path(args.output)
    if args.per_frame_ply:
        k = pipeline.export_frame_clouds(out_dir, subsample=config.subsample_factor)
        print(f"Saved {k} per-frame clouds to {out_dir / 'pointclouds'}")
    result = pipeline.reconstruct()
    if result[0] is None or len(result[0]) == 0:
        print("Reconstruction failed")
        return 0
    points, colors, poses = result
    out_dir.mkdir(parents=True, exist_ok=True)
    fileio.save_reconstruction(points, colors, out_dir / "reconstruction.ply")
    print(f"Saved {len(points)} points to {out_dir / 'reconstruction.ply'}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
