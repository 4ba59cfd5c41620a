"""User-written function blocks on the REFERENCE's own ABC (TEST INFRASTRUCTURE, build container only).

The reference's extension point is ``abstract_function_block`` (abstract_function_blocks.py:689-775): a user subclasses it,
gives ``num_inp`` / ``num_out`` / ``params`` and the two numba bodies, and composes the block with ``+``.  These blocks are
written by this repo (they are not reference code) and exist to pin pycamset_amd's counterpart of that extension point —
``function_blocks.device_function_block``, whose bodies are HIP device code — to what the reference's own code generator
makes of the same mathematics: ``make_golden.py --only round4`` imports this module from inside the temporary copy of the
reference (the generator reads a block's body with ``inspect.getsource`` and copies the imports of THIS file into the source it
writes, afb:245-262), runs ``make_full_loss_fn`` / ``make_jacobean`` on chains that contain the blocks and stores inputs and
outputs as ``tests/golden/user_*.npz``.  tests/test_gpu_dropin.py declares the same two blocks as device code.
"""
import numpy as np
from numba import njit
from pyCamSet.optimisation.abstract_function_blocks import abstract_function_block, key_type, param_type


class cam_scale(abstract_function_block):
    """One isotropic scale per camera, applied to the camera-frame point (between `projection` and `extrinsic3D`)."""
    num_inp = 3
    num_out = 3
    params = param_type(key_type.PER_CAM, 1)
    array_memory = 0

    @staticmethod
    @njit
    def compute_fun(params, inp, output, memory=0):
        output[0] = params[0] * inp[0]
        output[1] = params[0] * inp[1]
        output[2] = params[0] * inp[2]

    @staticmethod
    @njit
    def compute_jac(params, inp, output, memory=0):
        output[:12] = 0
        output[0] = inp[0]
        output[4] = inp[1]
        output[8] = inp[2]
        output[1] = params[0]
        output[6] = params[0]
        output[11] = params[0]


class division_projection(abstract_function_block):
    """Pinhole with the one-parameter division model of lens distortion, params = [fx, px, fy, py, lam]:
    (x, y) = (X / Z, Y / Z), d = 1 / (1 + lam (x^2 + y^2)), (u, v) = (fx x d + px, fy y d + py).  Replaces `projection`."""
    num_inp = 3
    num_out = 2
    params = param_type(key_type.PER_CAM, 5)
    array_memory = 0

    @staticmethod
    @njit
    def compute_fun(params, inp, output, memory=0):
        iz = 1 / inp[2]
        x = inp[0] * iz
        y = inp[1] * iz
        d = 1 / (1 + params[4] * (x * x + y * y))
        output[0] = params[0] * x * d + params[1]
        output[1] = params[2] * y * d + params[3]

    @staticmethod
    @njit
    def compute_jac(params, inp, output, memory=0):
        iz = 1 / inp[2]
        x = inp[0] * iz
        y = inp[1] * iz
        r2 = x * x + y * y
        d = 1 / (1 + params[4] * r2)
        d2 = d * d
        output[:16] = 0
        # row u: [fx, px, fy, py, lam | X, Y, Z]
        output[0] = x * d
        output[1] = 1
        output[4] = -params[0] * x * r2 * d2
        ux = params[0] * (d - 2 * params[4] * x * x * d2)
        uy = -2 * params[0] * params[4] * x * y * d2
        output[5] = ux * iz
        output[6] = uy * iz
        output[7] = -(x * ux + y * uy) * iz
        # row v
        output[8 + 2] = y * d
        output[8 + 3] = 1
        output[8 + 4] = -params[2] * y * r2 * d2
        vx = -2 * params[2] * params[4] * x * y * d2
        vy = params[2] * (d - 2 * params[4] * y * y * d2)
        output[8 + 5] = vx * iz
        output[8 + 6] = vy * iz
        output[8 + 7] = -(x * vx + y * vy) * iz


# the reference declares its shipped blocks' bodies with this signature string (function_block_implementations.py:11); a templated
# source written after the pattern of `template_points` (fbi:188-211) uses it too
ftemplate = "void(float64[::1],float64[::1],float64[::1],float64[::1])"


class board_flex(abstract_function_block):
    """A TEMPLATED user source (round 5): the reference hands `template[key]` to whatever block sits last when that block says
    ``template = True`` (afb:138, afb:374-375, afb:582).  One flex model of the calibration board per image, params =
    [sx, sy, tx, ty, k]:  out = [sx X + tx, sy Y + ty, Z + k (X^2 + Y^2)]  with (X, Y, Z) the template point."""
    template = True
    num_inp = 0
    num_out = 3
    params = param_type(key_type.PER_IMG, 5)
    array_memory = 0

    @staticmethod
    @njit(ftemplate, cache=True)
    def compute_fun(params, inp, output, memory):
        output[0] = params[0] * inp[0] + params[2]
        output[1] = params[1] * inp[1] + params[3]
        output[2] = inp[2] + params[4] * (inp[0] * inp[0] + inp[1] * inp[1])

    @staticmethod
    @njit(ftemplate, cache=True)
    def compute_jac(params, inp, output, memory):
        output[:15] = 0
        output[0] = inp[0]
        output[2] = 1
        output[5 + 1] = inp[1]
        output[5 + 3] = 1
        output[10 + 4] = inp[0] * inp[0] + inp[1] * inp[1]
