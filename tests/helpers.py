"""Shared helpers for the parity tests."""
import numpy as np


def copy_params(src, dst_cls):
    """Copy a ctypes Params struct between the oracle's and the library's (identically laid out) classes."""
    dst = dst_cls()
    for name, _ in src._fields_:
        if name == "label_score":
            for i in range(32):
                dst.label_score[i] = src.label_score[i]
        else:
            setattr(dst, name, getattr(src, name))
    return dst


def pose_err(Ta, Tb):
    """(max |rot diff| rad with wrap, max |trans diff| m)"""
    d = np.asarray(Ta, np.float64) - np.asarray(Tb, np.float64)
    rot = np.abs((d[..., :3] + np.pi) % (2 * np.pi) - np.pi).max()
    return float(rot), float(np.abs(d[..., 3:]).max())
