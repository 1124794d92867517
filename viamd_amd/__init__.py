"""viamd_amd — MI355X-native backend of VIAMD/mdlib's per-frame property evaluation (rdf / sdf / distance*).

The compute path is libviamd_amd.so (hand-written HIP for gfx950, C ABI in include/vmd_eval.h); this package is
the host-side mirror of the md_script evaluation interface.  Importing the package does not load the library;
the first object that needs it does, and fails loudly if it is not built (there is no CPU fallback).
"""
from ._lib import (DIST_COM, DIST_MAX, DIST_MIN, DIST_PAIR, FLAG_DISTRIBUTION, FLAG_TEMPORAL, FLAG_VOLUME, PBC_ALL,
                   RDF_NUM_BINS, VOLUME_DIM, VmdLib, default_lib)
from .eval import (MolSystem, PropertyDataView, ScriptEval, ScriptIR, VmdError, compute_histogram_masked,
                   downsample_histogram, make_unitcell)
from .dcd import DcdTrajectory, write_dcd
from .texttraj import TextTrajectory
from .xdr import CompressedDeviceTrajectory, XdrTrajectory, write_trr, write_xtc
from .trajectory import DeviceTrajectory, HostTrajectory, PinnedHostTrajectory

__all__ = ["ScriptIR", "ScriptEval", "MolSystem", "HostTrajectory", "DeviceTrajectory", "PinnedHostTrajectory", "DcdTrajectory", "write_dcd", "TextTrajectory",
           "XdrTrajectory", "CompressedDeviceTrajectory", "write_xtc", "write_trr",
           "PropertyDataView", "VmdError",
           "make_unitcell", "downsample_histogram", "compute_histogram_masked", "VmdLib", "default_lib"]
