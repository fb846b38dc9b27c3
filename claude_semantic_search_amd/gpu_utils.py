"""ROCm-aware device policy for the embed-and-search path.

Mirrors the decision inputs of the reference's ``src/gpu_utils.py`` (same
function and field names so its callers keep working) but asks the right
question on an MI355X: the reference labels a ROCm GPU "cuda", finds no
faiss-gpu (CUDA only) and keeps the index on the CPU (``src/gpu_utils.py:250-253``,
``src/storage.py:240-244``).  Here "faiss_gpu_available" means "libcss_hip.so
can see a HIP device", which is what runs the flat index.

Reference lines followed:
  * ``GPUCapability`` fields ............................ src/gpu_utils.py:17-29
  * ``estimate_gpu_memory_requirements`` ................ src/gpu_utils.py:142-166
  * ``calculate_optimal_batch_size`` (clamp 8..256) ..... src/gpu_utils.py:169-192
  * ``assess_gpu_capability`` verdict rules ............. src/gpu_utils.py:195-267
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple

logger = logging.getLogger(__name__)


@dataclass
class GPUCapability:
    torch_cuda_available: bool = False
    faiss_gpu_available: bool = False
    gpu_count: int = 0
    gpu_memory_total: Optional[int] = None
    gpu_memory_free: Optional[int] = None
    gpu_names: list = None
    recommended_batch_size: int = 16
    can_use_gpu: bool = False
    status_message: str = ""


def detect_hip_devices() -> Tuple[bool, Dict[str, Any]]:
    """Enumerate HIP devices through libcss_hip (name, CUs, HBM total/free, arch)."""
    try:
        from . import _native as nat

        n = nat.device_count()
        if n <= 0:
            return False, {"reason": "no HIP device visible to libcss_hip"}
        devices = []
        for i in range(n):
            info = nat.device_info(i)
            devices.append(
                {
                    "id": i,
                    "name": info["name"],
                    "gcn_arch": info["gcn_arch"],
                    "compute_units": info["compute_units"],
                    "memory_total": info["hbm_total_bytes"],
                    "memory_free": info["hbm_free_bytes"],
                    "memory_total_gb": info["hbm_total_bytes"] / (1024**3),
                    "memory_free_gb": info["hbm_free_bytes"] / (1024**3),
                }
            )
        return True, {"backend": "hip", "gpu_count": n, "devices": devices}
    except Exception as e:  # library not built, runtime missing, ...
        return False, {"reason": f"libcss_hip unavailable: {e}"}


def detect_torch_gpu() -> Tuple[bool, Dict[str, Any]]:
    """PyTorch-ROCm reports HIP devices through the ``torch.cuda`` namespace."""
    try:
        import torch

        if torch.cuda.is_available() and torch.cuda.device_count() > 0:
            devs = []
            for i in range(torch.cuda.device_count()):
                p = torch.cuda.get_device_properties(i)
                free = p.total_memory - torch.cuda.memory_allocated(i)
                devs.append(
                    {
                        "id": i,
                        "name": p.name,
                        "memory_total": p.total_memory,
                        "memory_free": free,
                        "memory_total_gb": p.total_memory / (1024**3),
                        "memory_free_gb": free / (1024**3),
                        "compute_capability": f"{p.major}.{p.minor}",
                    }
                )
            backend = "hip" if getattr(torch.version, "hip", None) else "cuda"
            return True, {"backend": backend, "gpu_count": len(devs), "devices": devs}
        return False, {"reason": "torch.cuda.is_available() is False"}
    except ImportError as e:
        return False, {"reason": f"PyTorch not installed: {e}"}
    except Exception as e:
        return False, {"reason": f"Error detecting PyTorch GPU: {e}"}


def detect_faiss_gpu() -> Tuple[bool, Dict[str, Any]]:
    """Name kept for the reference's callers: is the device flat index usable?"""
    ok, info = detect_hip_devices()
    if ok:
        return True, {"gpu_count": info["gpu_count"], "faiss_version": "css_hip"}
    return False, info


def estimate_gpu_memory_requirements(num_chunks: int, embedding_dim: int = 768) -> Dict[str, float]:
    # same model as the reference (GB): fp32 index + ~0.5 GB model + 10 % working + 20 % margin
    index_gb = (num_chunks * embedding_dim * 4) / (1024**3)
    model_gb = 0.5
    working_gb = (index_gb + model_gb) * 0.1
    total = index_gb + model_gb + working_gb
    return {
        "index_memory_gb": index_gb,
        "model_memory_gb": model_gb,
        "working_memory_gb": working_gb,
        "total_memory_gb": total,
        "recommended_gpu_memory_gb": total * 1.2,
    }


def calculate_optimal_batch_size(available_memory_gb: float, embedding_dim: int = 768, backend: str = "cuda") -> int:
    """``clamp(int((free_GB - 1) / (dim*16 / 2^30)), 8, 256)`` (64 for "mps").

    On any GPU with more than ~1 GB free this is 256 -- the "batch 256" of
    BASELINE.json comes from here (``src/gpu_utils.py:181-190``)."""
    working = available_memory_gb - 1.0
    if working <= 0:
        return 8
    per_item_gb = (embedding_dim * 4 * 4) / (1024**3)
    cap = 64 if backend == "mps" else 256
    return max(8, min(int(working / per_item_gb), cap))


def assess_gpu_capability(target_chunks: int = 10000, embedding_dim: int = 768) -> GPUCapability:
    cap = GPUCapability()
    hip_ok, hip_info = detect_hip_devices()
    torch_ok, _ = detect_torch_gpu()
    cap.torch_cuda_available = torch_ok
    cap.faiss_gpu_available = hip_ok
    if not hip_ok:
        cap.can_use_gpu = False
        cap.status_message = f"❌ GPU unavailable: {hip_info.get('reason', 'no HIP device')}"
        return cap
    primary = hip_info["devices"][0]
    cap.gpu_count = hip_info["gpu_count"]
    cap.gpu_memory_total = primary["memory_total"]
    cap.gpu_memory_free = primary["memory_free"]
    cap.gpu_names = [d["name"] for d in hip_info["devices"]]
    free_gb = primary["memory_free_gb"]
    cap.recommended_batch_size = calculate_optimal_batch_size(free_gb, embedding_dim)
    need = estimate_gpu_memory_requirements(target_chunks, embedding_dim)["recommended_gpu_memory_gb"]
    if free_gb >= need:
        cap.can_use_gpu = True
        cap.status_message = (
            f"✅ HIP GPU ready: {primary['name']} ({primary['gcn_arch']}, {primary['compute_units']} CUs; "
            f"Free: {free_gb:.1f}GB, Required: {need:.1f}GB)"
        )
    else:
        cap.can_use_gpu = False
        cap.status_message = f"⚠️ Insufficient GPU memory (Free: {free_gb:.1f}GB, Required: {need:.1f}GB)"
    return cap


def log_gpu_status(capability: GPUCapability, log: Optional[logging.Logger] = None) -> None:
    log = log or logger
    log.info(f"GPU Status: {capability.status_message}")
    if capability.gpu_names:
        total_gb = (capability.gpu_memory_total or 0) / (1024**3)
        for i, name in enumerate(capability.gpu_names):
            log.info(f"  GPU {i}: {name} ({total_gb:.1f}GB)")
    log.info("Flat index on device: " + ("✅ Available" if capability.faiss_gpu_available else "❌ Unavailable"))
    if capability.can_use_gpu:
        log.info(f"Recommended batch size: {capability.recommended_batch_size}")


def quick_gpu_check() -> bool:
    return assess_gpu_capability().can_use_gpu


def get_gpu_summary() -> str:
    cap = assess_gpu_capability()
    if not cap.can_use_gpu:
        return "❌ GPU: Unavailable"
    name = cap.gpu_names[0] if cap.gpu_names else "Available"
    mem = f" ({cap.gpu_memory_free / (1024**3):.1f}GB free)" if cap.gpu_memory_free else ""
    return f"✅ GPU: {name}{mem}"
