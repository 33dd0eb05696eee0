"""Depth from a LOCAL Depth-Anything checkpoint, for the DER-compatible driver.

depth_enhanced_reconstruction.py:87-171 (`DepthEstimator`) loads `depth-anything/Depth-Anything-V2-Large-hf` BY NAME from the
HuggingFace hub and runs it per image: BGR -> RGB -> image processor -> model -> bicubic interpolation back to the image size
(DER:139-165).  The network itself is out of scope of this back end and stays upstream PyTorch-ROCm (north star); what is kept
here is the seam: the same call sequence on the upstream `transformers` classes, with the checkpoint read from a directory
(`local_files_only=True`, no network), and the result left ON THE GPU so that it goes straight into a frame slot
(tl3d_upload_frame accepts device pointers) instead of through `.cpu().numpy()`.
"""
from __future__ import annotations

import os

import numpy as np


class LocalDepthEstimator:
    def __init__(self, model_dir: str, device: int = 0):
        model_dir = os.fspath(model_dir)
        if not os.path.isdir(model_dir) or not os.path.exists(os.path.join(model_dir, "config.json")):
            raise FileNotFoundError(
                f"--depth-model: {model_dir!r} is not a local checkpoint directory (expected config.json, the weights and "
                "preprocessor_config.json, as written by save_pretrained). The reference downloads "
                "'depth-anything/Depth-Anything-V2-Large-hf' by name (depth_enhanced_reconstruction.py:114-118); this driver "
                "never touches the network: download the checkpoint elsewhere and pass its directory.")
        import torch
        from transformers import AutoModelForDepthEstimation
        self.torch = torch
        self.device = torch.device("cuda", int(device))
        self.model = AutoModelForDepthEstimation.from_pretrained(model_dir, local_files_only=True).to(self.device).eval()
        try:                                         # the reference's class (DER:116); needs torchvision in transformers 5
            from transformers import AutoImageProcessor
            self.processor = AutoImageProcessor.from_pretrained(model_dir, local_files_only=True)
        except ImportError:                          # same preprocessing, PIL back end (no torchvision in this image)
            from transformers import DPTImageProcessorPil
            self.processor = DPTImageProcessorPil.from_pretrained(model_dir, local_files_only=True)
        print(f"Depth model loaded on {self.device} from {model_dir}")

    def estimate(self, image_bgr: np.ndarray):
        """Relative depth of one BGR image as a float32 [H, W] tensor on the GPU (DER:139-165)."""
        from PIL import Image as PILImage
        torch = self.torch
        h, w = image_bgr.shape[:2]
        pil = PILImage.fromarray(np.ascontiguousarray(image_bgr[..., ::-1]))
        inputs = self.processor(images=pil, return_tensors="pt").to(self.device)
        with torch.no_grad():
            pred = self.model(**inputs).predicted_depth
        depth = torch.nn.functional.interpolate(pred.unsqueeze(1), size=(h, w), mode="bicubic", align_corners=False).squeeze(1).squeeze(0)
        depth = depth.to(torch.float32).contiguous()
        torch.cuda.current_stream(self.device).synchronize()      # produced on torch's stream, consumed on the library's
        return depth

    def estimate_batch(self, images):
        return [self.estimate(img) for img in images]
