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
                ('view_pe', ctypes.c_int32), ('sdf_index_exact', ctypes.c_int32)]


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
    L.pp_abi_version.restype = ctypes.c_int
    want = header_abi_version()
    if L.pp_abi_version() != want:           # signatures differ between ABI versions: never call into a mismatching build
        raise PoseProbeError(f'{SO_PATH} reports ABI {L.pp_abi_version()}, include/poseprobe_hip.h declares {want}: '
                             'rebuild with `python -m poseprobe_amd.build_ext`')
    for name, args in parse_header().items():
        fn = getattr(L, name)           # raises AttributeError if a declared symbol is not exported
        fn.restype = ctypes.c_int
        fn.argtypes = [ct for ct, _ in args]
    _lib = L
    return L


def header_abi_version(path=HEADER):
    return int(re.search(r'#define\s+PP_ABI_VERSION\s+(\d+)', open(path).read()).group(1))


# ---- options ------------------------------------------------------------------------------------------------------------
# The library has no process-wide state: options are fields of a caller-owned pp_context handed to each call
# (include/poseprobe_hip.h).  The Python host keeps ONE default context per process for callers that do not bring their own;
# PP_<NAME>=<int> in the host's environment seeds it once.  set_option / get_option below act on that host-side default
# context (A/B scripts, tests); engines built with `options=...` own a private context and are unaffected by it.
OPTION_NAMES = ('mlp_fused', 'wgrad_split', 'grid_chunks', 'nerf_split', 'nerf_split_tn', 'nerf_bitmask', 'nerf_gemm_wgs',
                'nerf_tn_ch', 'nerf_tn_split_wgs', 'nerf_tn_wgs', 'nerf_bn', 'nerf_planes', 'mlp_split', 'nerf_tn256', 'mlp_wgs',
                'wgrad_side_wgs', 'side_stream', 'nerf_chain', 'nerf_chain_nw', 'nerf_chain_head', 'nerf_tn_tr')


class Context:
    """Caller-owned pp_context: option values (+ the auxiliary stream the library creates in it on first use)."""

    def __init__(self, **options):
        self.handle = ctypes.c_void_p()
        check(lib().pp_context_create(ctypes.byref(self.handle)), 'pp_context_create')
        for k, v in options.items():
            self.set(k, v)

    def set(self, name, value):
        check(lib().pp_context_set_option(self.handle, name.encode(), int(value)), 'pp_context_set_option')

    def get(self, name):
        v = ctypes.c_int32()
        check(lib().pp_context_get_option(self.handle, name.encode(), ctypes.byref(v)), 'pp_context_get_option')
        return v.value

    def options(self):
        return {k: self.get(k) for k in OPTION_NAMES}

    def __del__(self):
        try:
            if _lib is not None and self.handle:
                _lib.pp_context_destroy(self.handle)
        except Exception:
            pass


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        c = Context()
        for name in OPTION_NAMES:
            v = os.environ.get('PP_' + name.upper())
            if v not in (None, ''):
                c.set(name, int(v))
        if os.environ.get('PP_SIDE_STREAM') in ('1', '2'):
            c.set('side_stream', int(os.environ['PP_SIDE_STREAM']))
        _default_ctx = c
    return _default_ctx


def handle(ctx):
    """ctypes handle of `ctx` (a Context, or None = the host's default context)."""
    return (default_context() if ctx is None else ctx).handle


def set_option(name, value):
    default_context().set(name, value)


def get_option(name):
    return default_context().get(name)


def library_default(name):
    v = ctypes.c_int32()
    check(lib().pp_context_get_option(None, name.encode(), ctypes.byref(v)), 'pp_context_get_option')
    return v.value


def check(status, name):
    if status != 0:
        raise PoseProbeError(f'{name} failed ({status}): {lib().pp_last_error().decode()}')


def call(name, *args):
    check(getattr(lib(), name)(*args), name)
