"""bc_mask encoding (reference xlb/cell_type.py:9-11): 0 = fluid, 1..253 = registered boundary-condition ids,
254 / 255 reserved by the reference for multires "simple fluid voxels" / solid voxels."""

BC_NONE, BC_SFV, BC_SOLID = 0, 254, 255
