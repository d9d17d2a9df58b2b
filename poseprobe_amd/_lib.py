"""ctypes binding of libposeprobe_hip.so.  Prototypes are derived from include/poseprobe_hip.h so that the
Python side can never drift from the C ABI.  There is NO fallback: if the library is missing the import of
any op fails loudly (the product path never routes around the HIP kernels)."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, '..', 'include', 'poseprobe_hip.h')
SO_PATH = os.path.join(HERE, 'libposeprobe_hip.so')


class pp_scene(ctypes.Structure):
    _fields_ = [('xyz_min', ctypes.c_float * 3), ('xyz_max', ctypes.c_float * 3), ('size', ctypes.c_int32 * 3),
                ('voxel_size', ctypes.c_float), ('stepsize', ctypes.c_float), ('near_clip', ctypes.c_float),
                ('far_clip', ctypes.c_float), ('bg', ctypes.c_float), ('n_samples', ctypes.c_int32),
                ('out_range', ctypes.c_float), ('k0_dim', ctypes.c_int32), ('pos_pe', ctypes.c_int32),
                ('view_pe', ctypes.c_int32)]


def parse_header(path=HEADER):
    """-> {name: [(ctype, argname), ...]} for every `int pp_*(...)` prototype in the header."""
    text = open(path).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    protos = {}
    for m in re.finditer(r'\bint\s+(pp_\w+)\s*\(([^;]*?)\)\s*;', text, flags=re.S):
        name, args = m.group(1), m.group(2)
        parsed = []
        for a in args.split(','):
            a = ' '.join(a.split())
            if a in ('void', ''):
                continue
            argname = re.findall(r'(\w+)$', a)[0]
            if 'pp_scene' in a:
                ct = ctypes.POINTER(pp_scene)
            elif 'char*' in a.replace(' *', '*'):
                ct = ctypes.c_char_p
            elif '*' in a:
                ct = ctypes.c_void_p
            elif 'uint8_t' in a and '*' not in a:
                ct = ctypes.c_uint8
            elif 'int64_t' in a:
                ct = ctypes.c_int64
            elif 'int32_t' in a or a.startswith('int '):
                ct = ctypes.c_int32
            elif 'float' in a:
                ct = ctypes.c_float
            else:
                raise ValueError(f'unhandled argument "{a}" in {name}')
            parsed.append((ct, argname))
        protos[name] = parsed
    return protos


class PoseProbeError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise PoseProbeError(f'{SO_PATH} is missing: run `python -m poseprobe_amd.build_ext` '
                             '(there is no CPU / eager fallback for the hot path)')
    L = ctypes.CDLL(SO_PATH)
    L.pp_last_error.restype = ctypes.c_char_p
    for name, args in parse_header().items():
        fn = getattr(L, name)           # raises AttributeError if a declared symbol is not exported
        fn.restype = ctypes.c_int
        fn.argtypes = [ct for ct, _ in args]
    _lib = L
    _options_from_environment(L)
    return L


# PP_<NAME>=<int> in the environment of the HOST process is applied once, right after loading, through pp_set_option: the
# switches are the host's, the library itself has no hidden state read from the environment.
OPTION_NAMES = ('mlp_fused', 'wgrad_split', 'grid_chunks', 'nerf_split', 'nerf_split_tn', 'nerf_bitmask', 'nerf_gemm_wgs',
                'nerf_tn_ch', 'nerf_tn_split_wgs', 'nerf_tn_wgs', 'nerf_bn', 'nerf_planes', 'sdf_index_exact', 'mlp_split', 'nerf_tn256', 'mlp_wgs', 'wgrad_side_wgs')


def _options_from_environment(L):
    for name in OPTION_NAMES:
        v = os.environ.get('PP_' + name.upper())
        if v not in (None, ''):
            check(L.pp_set_option(name.encode(), int(v)), 'pp_set_option')


def set_option(name, value):
    check(lib().pp_set_option(name.encode(), int(value)), 'pp_set_option')


def get_option(name):
    v = ctypes.c_int32()
    check(lib().pp_get_option(name.encode(), ctypes.byref(v)), 'pp_get_option')
    return v.value


def check(status, name):
    if status != 0:
        raise PoseProbeError(f'{name} failed ({status}): {lib().pp_last_error().decode()}')


def call(name, *args):
    check(getattr(lib(), name)(*args), name)
