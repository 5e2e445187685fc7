"""bc_mask encoding (reference xlb/cell_type.py:9-11): 0 = fluid, 1..253 = registered
boundary-condition ids, 254/255 reserved by the reference for multires / solid voxels."""

BC_NONE = 0
BC_SFV = 254
BC_SOLID = 255
