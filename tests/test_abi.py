"""CPU: the C-ABI library builds, loads, and exports every symbol include/*.h declares; without a
GPU the product refuses to run (no CPU fallback)."""
import ctypes
import glob
import os
import re

import pytest

import _harness as H

LIB = os.path.join(H.ROOT, "soft-rendering-toolsets_amd", "lib", "libsrt_hip.so")


def declared_symbols():
    syms = set()
    for hdr in glob.glob(os.path.join(H.ROOT, "include", "*.h")):
        text = open(hdr).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms |= set(re.findall(r"\b(srt_[a-z0-9_]+)\s*\(", text))
    return sorted(syms)


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__ as g

        g.build()
    return ctypes.CDLL(LIB)


def test_headers_declare_the_boundary():
    syms = declared_symbols()
    for s in ("srt_raster_create", "srt_raster_set_target", "srt_raster_clear", "srt_raster_submit",
              "srt_raster_resolve", "srt_raster_destroy", "srt_last_error"):
        assert s in syms


def test_library_exports_every_declared_symbol(lib):
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_headers_cite_reference_lines():
    for hdr in glob.glob(os.path.join(H.ROOT, "include", "*.h")):
        text = open(hdr).read()
        assert re.search(r"\.(cpp|h|inl):\d+", text), f"{hdr} cites no reference file:line"


def test_prim_record_layout():
    import srt_amd

    assert srt_amd.PRIM_DTYPE.itemsize == 48
    assert srt_amd.PRIM_DTYPE.fields["v"][1] == 8 and srt_amd.PRIM_DTYPE.fields["rgba"][1] == 32
    assert H.PRIM_DTYPE == srt_amd.PRIM_DTYPE


def test_no_cpu_fallback_without_device(lib):
    """On a box without a GPU the product must fail loudly, not compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import srt_amd

    with pytest.raises(srt_amd.SrtError) as e:
        srt_amd.SoftwareRenderer()
    assert e.value.status == -2  # SRT_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_reference_the_oracle():
    """Nothing under the product package may import, link or dlopen oracle/ ."""
    pkg = os.path.join(H.ROOT, "soft-rendering-toolsets_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".c", "Makefile")):
                text = open(os.path.join(root, f), errors="replace").read()
                assert "liboracle" not in text and "oracle/" not in text and "_ref/" not in text, os.path.join(root, f)
