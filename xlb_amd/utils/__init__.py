"""Host-side helpers around the stepper — the callers' side of the path (reference: xlb/utils/utils.py): image and VTK output, STL
input, geometry helpers, unit conversion.  Pure Python / NumPy (matplotlib for PNGs); nothing here is on the per-step path."""

from .utils import (  # noqa: F401
    UnitConvertor,
    axangle2mat,
    downsample_field,
    load_stl,
    read_fields_vtk,
    rotate_geometry,
    save_fields_vtk,
    save_image,
    save_stl,
    voxelize_stl,
)
