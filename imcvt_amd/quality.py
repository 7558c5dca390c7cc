"""Decoder-free quality check of the H.265 path: the encoder hands back its reconstruction
(`img_rcon`, padded stride — src/HEVCe/HEVCe.c:1614,1627), so PSNR against the source needs no HEVC decoder.
Integer sums, one float division at the end."""
import math

import numpy as np


def crop_rcon(rcon: np.ndarray, h: int, w: int) -> np.ndarray:
    """The h x w picture inside a padded reconstruction (the padding replicates edge pixels, :1621-1626)."""
    return rcon[: min(h, rcon.shape[0]), : min(w, rcon.shape[1])]


def sse(img: np.ndarray, rcon: np.ndarray) -> int:
    a = img.astype(np.int64)
    b = crop_rcon(rcon, *img.shape).astype(np.int64)
    a = a[: b.shape[0], : b.shape[1]]                      # pictures beyond 8192 are cropped by the encoder (:1580-1581)
    return int(((a - b) ** 2).sum())


def psnr(img: np.ndarray, rcon: np.ndarray) -> float:
    """PSNR in dB of an 8-bit picture against its (padded) reconstruction; inf when identical."""
    e = sse(img, rcon)
    n = min(img.shape[0], rcon.shape[0]) * min(img.shape[1], rcon.shape[1])
    return math.inf if e == 0 else 10.0 * math.log10(255.0 * 255.0 * n / e)


def rd_point(img: np.ndarray, stream: bytes, rcon: np.ndarray) -> dict:
    """One rate-distortion point: bytes, bits per pixel and PSNR."""
    return {"bytes": len(stream), "bpp": 8.0 * len(stream) / img.size, "psnr_db": psnr(img, rcon)}
