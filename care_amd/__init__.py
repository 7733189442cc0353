"""care_amd - MI355X-native captioning forward path of yangbang18/CARE.

Public seam (mirrors the reference's two factories):
    from care_amd import get_framework, get_translator
"""
from .framework import get_framework  # noqa: F401
from .translator import get_translator  # noqa: F401

__all__ = ["get_framework", "get_translator"]
