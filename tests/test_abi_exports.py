"""The C-ABI library loads without a GPU and exports every entry point include/*.h declares (no compute call is made here); the
product header declares no test hook (round-4 verdict item 6: they live in contrack_hip_debug.h)."""
import os
import re

from contrack_amd import _native

INC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")


def declared(header):
    text = open(os.path.join(INC, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)                         # comments mention functions too
    return sorted(set(re.findall(r"^\s*(?:const\s+)?(?:int|void|double|char)\s*\*?\s*(ctk_\w+)\s*\(", text, flags=re.M)))


def test_every_declared_entry_point_is_exported():
    lib = _native.lib()
    prod, dbg = declared("contrack_hip.h"), declared("contrack_hip_debug.h")
    assert len(prod) > 60 and len(dbg) >= 20
    missing = [n for n in prod + dbg if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(set(_native.EXPORTS) - set(prod) - set(dbg)) == []          # the binding's list names nothing undeclared


def test_product_header_declares_no_test_hook():
    prod = declared("contrack_hip.h")
    assert not [n for n in prod if n.startswith("ctk_debug_")]
    for n in ("ctk_create", "ctk_destroy", "ctk_last_error", "ctk_track_f32", "ctk_track_f32_dev", "ctk_track_sharded_f32_dev", "ctk_track_stream_cb",
              "ctk_lifecycle_f32", "ctk_anom_f32", "ctk_comm_init_rccl"):
        assert n in prod, n
