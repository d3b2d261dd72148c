"""Device-side engine: thin Python over the C ABI (include/sdsm.h).

PyTorch-ROCm tensors own every device buffer; their ``data_ptr()`` and the current stream's handle are passed
to libsdsm_hip.so.  Nothing here computes on the CPU: if the HIP library or a GPU is missing, construction
fails loudly.
"""
import ctypes as C
import itertools
import threading

import numpy as np
import torch

from . import _capi


def _require_gpu():
    if not torch.cuda.is_available():
        raise _capi.SdsmError('no HIP device available: the DSM solve path has no CPU fallback')


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def preprocess(g_raw, sigma1=np.sqrt(2), sigma2=40, offset_clip=3, lower_clip_mean=False, device=None, return_tensor=False):
    """Offset-intensity map ``y`` (reference: Preprocessing.process, superdsm/preprocess.py:39-68)."""
    _require_gpu()
    L = _capi.lib()
    dev = torch.device(device if device is not None else 'cuda')
    g = torch.as_tensor(np.ascontiguousarray(g_raw, dtype=np.float64)).to(dev) if not torch.is_tensor(g_raw) else g_raw.to(dev, torch.float64).contiguous()
    H, W = g.shape
    y = torch.empty_like(g)
    nbytes = L.sdsm_preprocess_workspace_bytes(H, W, float(sigma1), float(sigma2))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _capi.check(L.sdsm_preprocess(_ptr(g), H, W, float(sigma1), float(sigma2), float(offset_clip), int(bool(lower_clip_mean)),
                                      _ptr(y), _ptr(ws), nbytes, _stream()), 'sdsm_preprocess')
    return y if return_tensor else y.cpu().numpy()


class DeviceImage:
    """Per-image device state: y, atoms, the candidate-independent validity mask and per-atom extents.

    Replaces the per-candidate O(H*W) work of Object.get_cvxprog_region (superdsm/objects.py:95-128): the EDT
    term does not depend on the candidate and is evaluated once.
    """

    def __init__(self, y, y_mask, atoms, background_margin, device=None):
        _require_gpu()
        L = _capi.lib()
        self.device = torch.device(device if device is not None else 'cuda')
        as_dev = lambda a, dt: (a.to(self.device, dt).contiguous() if torch.is_tensor(a) else torch.as_tensor(np.ascontiguousarray(a, dtype=dt)).to(self.device))
        self.y = as_dev(y, torch.float64 if torch.is_tensor(y) else np.float64)
        self.atoms = as_dev(atoms, torch.int32 if torch.is_tensor(atoms) else np.int32)
        self.y_mask = None if y_mask is None else as_dev(y_mask, torch.uint8 if torch.is_tensor(y_mask) else np.uint8)
        self.H, self.W = (int(v) for v in self.y.shape)
        assert tuple(self.atoms.shape) == (self.H, self.W)
        self.background_margin = float(background_margin)
        self.n_atoms = int(self.atoms.max().item()) if self.atoms.numel() else 0
        self.valid = torch.empty((self.H, self.W), dtype=torch.uint8, device=self.device)
        stats = torch.empty(((self.n_atoms + 1) * _capi.ATOM_STATS_STRIDE,), dtype=torch.int32, device=self.device)
        nbytes = L.sdsm_image_workspace_bytes(self.H, self.W)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _capi.check(L.sdsm_image_prepare(_ptr(self.y), _ptr(self.y_mask), _ptr(self.atoms), self.H, self.W, self.background_margin,
                                             self.n_atoms, _ptr(self.valid), _ptr(stats), _ptr(ws), nbytes, _stream()), 'sdsm_image_prepare')
        self.atom_stats = np.ascontiguousarray(stats.cpu().numpy())       # host copy used by the planner (synchronises)


_PINNED = {}                       # (device type, index, stream) -> (records staging, masks staging): pinned, grown on demand, shared by the batches of a stream
_PINNED_LOCK = threading.Lock()


class PackedFragments:
    """The bit-packed region-bbox masks of one batch, kept on the host: fragment ``i`` is unpacked when somebody looks at it
    (``Object.fg_fragment``) -- most candidates of a generation are pruned by their energy alone."""

    def __init__(self, records, mask_info, mask_offset, masks):
        self.records, self.mask_info, self.mask_offset, self.masks = records, mask_info, mask_offset, masks

    def get(self, i):
        return fragments_from_masks(self.records[i:i + 1], self.mask_info[i:i + 1], self.mask_offset[i:i + 1], self.masks)[0][1]


def _empty_fragment(records):
    return (records['fg_h'] <= 0) | np.isin(records['status'], (_capi.CAND_TRIVIAL, _capi.CAND_ERROR, _capi.CAND_GIVEN_UP))


def fragments_from_masks(records, mask_info, mask_offset, masks, select=None, lazy=False):
    """Foreground fragments (bool arrays, views into one buffer) and offsets from downloaded records and bit-packed
    region-bbox masks (objects.py:148-174); ``select`` (bool per candidate) skips the others (``(None, None)``).  ``lazy``: the fragment
    as ``(PackedFragments, index)`` -- what ``Object.fg_fragment`` turns into the array on first access; ``masks`` is copied (it may be
    a view of a staging buffer)."""
    L = _capi.lib()
    n = len(records)
    records = np.ascontiguousarray(records)
    mask_info = np.ascontiguousarray(mask_info[:n], np.int32)
    mask_offset = np.ascontiguousarray(mask_offset[:n], np.int64)
    masks = np.ascontiguousarray(masks, np.uint8)
    if lazy and select is None:
        empty = _empty_fragment(records)
        origin = np.stack([np.where(empty, 0, records['fg_r0']), np.where(empty, 0, records['fg_c0'])], axis=1).astype(int)
        src = PackedFragments(records.copy(), mask_info.copy(), mask_offset.copy(), masks.copy())      # (the inputs may be views of staging buffers)
        return [(o, (src, i)) for i, o in enumerate(origin)]
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    off = np.zeros(n + 1, np.int64)
    total = L.sdsm_unpack_fragments(ptr(records), ptr(mask_info), ptr(mask_offset), ptr(masks), n, None, ptr(off))
    if total < 0:
        raise _capi.SdsmError('sdsm_unpack_fragments: ' + L.sdsm_last_error().decode())
    buf = np.empty(max(total, 1), np.uint8)
    L.sdsm_unpack_fragments(ptr(records), ptr(mask_info), ptr(mask_offset), ptr(masks), n, ptr(buf), ptr(off))
    off[n] = total
    fb = buf.view(bool)
    empty = _empty_fragment(records)
    fh = np.where(empty, 1, records['fg_h']).tolist()
    fw = np.where(empty, 1, records['fg_w']).tolist()
    origin = np.stack([np.where(empty, 0, records['fg_r0']), np.where(empty, 0, records['fg_c0'])], axis=1).astype(int)
    offs = off.tolist()
    out = []
    for i in range(n):
        if select is not None and not select[i]:
            out.append((None, None))
        else:
            out.append((origin[i], fb[offs[i]:offs[i + 1]].reshape(fh[i], fw[i])))
    return out


def plan_mask_boxes(image, footprints, dsm_cfg):
    """Region bounding boxes (r0, c0, h, w) of candidates, [n, 4] int32: host-only planning from the per-atom statistics
    (sdsm_plan_create + sdsm_plan_describe; no device access).  ``image`` needs H, W, n_atoms, atom_stats, background_margin."""
    L = _capi.lib()
    n = len(footprints)
    offs = np.zeros(n + 1, np.int32)
    np.cumsum(np.fromiter(map(len, footprints), np.int64, n), out=offs[1:])
    labels = np.fromiter(itertools.chain.from_iterable(footprints), np.int32, int(offs[-1]))
    cfg = _capi.make_config(dict(dsm_cfg, background_margin=image.background_margin))
    stats = np.ascontiguousarray(image.atom_stats, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    plan = L.sdsm_plan_create(int(image.H), int(image.W), int(image.n_atoms), p(stats), C.byref(cfg), n, p(offs), p(labels))
    if not plan:
        raise _capi.SdsmError('sdsm_plan_create failed: ' + L.sdsm_last_error().decode())
    info = np.zeros((max(n, 1), 4), np.int32)
    _capi.check(L.sdsm_plan_describe(plan, p(info), None, None), 'sdsm_plan_describe')
    L.sdsm_plan_destroy(plan)
    return info[:n]


class Batch:
    """One ``compute_objects`` call: plan, workspace, launch, results.  ``image`` may be a list of :class:`DeviceImage` (a plan
    over several images, sdsm_plan_create_multi): ``image_of[i]`` then names the image of candidate ``i``."""

    def __init__(self, image, footprints, dsm_cfg, want_xi=False, latency_mode=False, image_of=None, mode=None):
        """mode: 0 throughput (default), 1 latency (``latency_mode=True``: the largest regions get a group of 512-thread
        workgroups, shortest wall clock of ONE batch), 2 no workgroup groups (sdsm_plan_set_latency_mode)."""
        L = _capi.lib()
        self.images = list(image) if isinstance(image, (list, tuple)) else [image]
        self.image = self.images[0]
        assert 1 <= len(self.images) <= 16, 'a plan covers 1 .. 16 images'
        self.n = len(footprints)
        lens = np.fromiter(map(len, footprints), np.int64, self.n)
        offs = np.zeros(self.n + 1, np.int32)
        np.cumsum(lens, out=offs[1:])
        labels = np.fromiter(itertools.chain.from_iterable(footprints), np.int32, int(offs[-1]))
        self.image_of = None if image_of is None else np.ascontiguousarray(image_of, np.int32)
        assert len(self.images) == 1 or (self.image_of is not None and len(self.image_of) == self.n)
        margins = {im.background_margin for im in self.images}
        assert len(margins) == 1, 'all images of a plan share the hyper-parameters'
        self.cfg = _capi.make_config(dict(dsm_cfg, background_margin=self.image.background_margin))
        ni = len(self.images)
        Hs = np.array([im.H for im in self.images], np.int32)
        Ws = np.array([im.W for im in self.images], np.int32)
        nas = np.array([im.n_atoms for im in self.images], np.int32)
        stats = (C.c_void_p * ni)(*[im.atom_stats.ctypes.data for im in self.images])
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.plan = L.sdsm_plan_create_multi(ni, p(Hs), p(Ws), p(nas), stats, C.byref(self.cfg), self.n, p(offs), p(labels),
                                             p(self.image_of) if self.image_of is not None else None)
        if not self.plan:
            raise _capi.SdsmError('sdsm_plan_create failed: ' + L.sdsm_last_error().decode())
        self.mode = (1 if latency_mode else 0) if mode is None else int(mode)
        if self.mode:
            _capi.check(L.sdsm_plan_set_latency_mode(self.plan, self.mode), 'sdsm_plan_set_latency_mode')
        dev = self.image.device
        self.ws_bytes = L.sdsm_plan_workspace_bytes(self.plan)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.records_dev = torch.zeros(max(self.n, 1) * _capi.RECORD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        self.mask_bytes = L.sdsm_plan_mask_bytes(self.plan)
        self.masks_dev = torch.empty(self.mask_bytes, dtype=torch.uint8, device=dev)
        self.xi_dev = torch.zeros(L.sdsm_plan_xi_count(self.plan), dtype=torch.float64, device=dev) if want_xi else None
        self.mask_info = np.zeros((max(self.n, 1), 4), np.int32)
        self.mask_offset = np.zeros(max(self.n, 1), np.int64)
        self.n_pixels = np.zeros(max(self.n, 1), np.int32)
        _capi.check(L.sdsm_plan_describe(self.plan, p(self.mask_info), p(self.mask_offset), p(self.n_pixels)), 'sdsm_plan_describe')
        self.total_pixels = L.sdsm_plan_total_pixels(self.plan)
        self._ptrs = tuple((C.c_void_p * ni)(*[getattr(im, k).data_ptr() for im in self.images]) for k in ('y', 'atoms', 'valid'))
        with torch.cuda.device(dev):
            _capi.check(L.sdsm_batch_upload(self.plan, _ptr(self.ws), self.ws_bytes, _stream()), 'sdsm_batch_upload')

    def __del__(self):
        plan = getattr(self, 'plan', None)
        if plan and _capi is not None and _capi._lib is not None:
            _capi._lib.sdsm_plan_destroy(plan)
            self.plan = None

    def launch(self):
        """Queues the setup and solve kernels on the current stream (asynchronous)."""
        L = _capi.lib()
        with torch.cuda.device(self.image.device):
            _capi.check(L.sdsm_batch_launch_multi(self.plan, self._ptrs[0], self._ptrs[1], self._ptrs[2], _ptr(self.ws), self.ws_bytes,
                                                  _ptr(self.records_dev), _ptr(self.masks_dev), _ptr(self.xi_dev), _stream()), 'sdsm_batch_launch_multi')

    def deform_counts(self):
        """Number of columns of every candidate's G~ (its grid points; -1: no solve), from a run of the setup kernel alone
        (sdsm_batch_deform_counts; synchronises): what a callable ``dsm/init`` is called with (objects.py:385-386)."""
        L = _capi.lib()
        m = np.full(max(self.n, 1), -1, np.int32)
        with torch.cuda.device(self.image.device):
            _capi.check(L.sdsm_batch_deform_counts(self.plan, self._ptrs[0], self._ptrs[1], self._ptrs[2], _ptr(self.ws), self.ws_bytes,
                                                   m.ctypes.data_as(C.c_void_p), _stream()), 'sdsm_batch_deform_counts')
        return m[:self.n]

    def set_start(self, params):
        """Starting points of the DSM solves of the following launches (sdsm_plan_set_start): ``params[i]`` = theta (6,
        full-image-normalised) + xi (M) of candidate i, or None for a candidate without a solve.  Only for plans with
        ``init`` other than ``'elliptical'``."""
        L = _capi.lib()
        xo = self.xi_offsets()
        buf = np.zeros(L.sdsm_plan_eval_param_count(self.plan))
        for i, p in enumerate(params):
            if p is None:
                continue
            p = np.asarray(p, np.float64).ravel()
            buf[6 * i + xo[i]:6 * i + xo[i] + p.size] = p
        self.x0_dev = torch.from_numpy(buf).to(self.image.device)       # (kept alive by the batch)
        _capi.check(L.sdsm_plan_set_start(self.plan, _ptr(self.x0_dev)), 'sdsm_plan_set_start')

    def download(self):
        """Records and bit-packed masks to pinned host staging buffers: two asynchronous copies on the current stream, one
        synchronisation.  Returns (records structured array, masks uint8 array) -- VIEWS of the staging buffers, which are shared
        by all batches of the device AND STREAM and valid until the next download on it (copy what must live longer)."""
        dev = self.image.device
        key = (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)     # per stream: image groups in a pipeline download concurrently
        with _PINNED_LOCK:
            cur = _PINNED.get(key)
            need = (self.records_dev.numel(), self.masks_dev.numel())
            if cur is None or cur[0].numel() < need[0] or cur[1].numel() < need[1]:
                grow = lambda old, n: max(n, 2 * old) if old else max(n, 1 << 16)      # page-locking memory costs milliseconds: amortised
                cur = tuple(torch.empty(grow(0 if cur is None else cur[k].numel(), need[k]), dtype=torch.uint8).pin_memory() for k in range(2))
                _PINNED[key] = cur
        with torch.cuda.device(dev):
            cur[0][:need[0]].copy_(self.records_dev, non_blocking=True)
            cur[1][:need[1]].copy_(self.masks_dev, non_blocking=True)
            torch.cuda.current_stream().synchronize()
        return cur[0].numpy()[:need[0]].view(_capi.RECORD_DTYPE)[:self.n], cur[1].numpy()[:need[1]]

    def records(self):
        return self.records_dev.cpu().numpy().view(_capi.RECORD_DTYPE)[:self.n].copy()

    def xi_offsets(self):
        off = np.zeros(max(self.n, 1), np.int64)
        _capi.check(_capi.lib().sdsm_plan_xi_offsets(self.plan, off.ctypes.data_as(C.c_void_p)), 'sdsm_plan_xi_offsets')
        return off

    def evaluate(self, params):
        """Point evaluation for parity tests (sdsm_batch_eval): ``params[i]`` = theta (6, full-image-normalised) + xi (M) of
        candidate i.  Needs a previous :meth:`launch`.  Returns a list of dicts: psi (full evaluator), psi_value
        (value-only evaluator), grad (6 + M), hess_theta (6 x 6, symmetric)."""
        L = _capi.lib()
        xo = self.xi_offsets()
        buf = np.zeros(L.sdsm_plan_eval_param_count(self.plan))
        for i, p in enumerate(params):
            p = np.asarray(p, np.float64)
            buf[6 * i + xo[i]:6 * i + xo[i] + p.size] = p
        dev = self.image.device
        d_par = torch.from_numpy(buf).to(dev)
        d_out = torch.empty(L.sdsm_plan_eval_out_count(self.plan), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            _capi.check(L.sdsm_batch_eval(self.plan, _ptr(self.ws), self.ws_bytes, _ptr(d_par), _ptr(d_out), _stream()), 'sdsm_batch_eval')
        out = d_out.cpu().numpy()
        res = []
        for i, p in enumerate(params):
            m = len(p)
            H = np.zeros((6, 6))
            H[np.tril_indices(6)] = out[2 * self.n + 21 * i:2 * self.n + 21 * i + 21]
            H = H + np.tril(H, -1).T
            g0 = 23 * self.n + 6 * i + xo[i]
            res.append(dict(psi=float(out[2 * i]), psi_value=float(out[2 * i + 1]), grad=out[g0:g0 + m].copy(), hess_theta=H))
        return res

    def inspect_states(self):
        """The per-candidate state words of the setup kernel (M, status, envelope size, runs) without the crops / rows of :meth:`inspect`."""
        lay = np.zeros(16, np.int64)
        _capi.check(_capi.lib().sdsm_plan_layout(self.plan, lay.ctypes.data_as(C.c_void_p)), 'sdsm_plan_layout')
        ssz = int(lay[14])
        state = self.ws[int(lay[1]):int(lay[1]) + ssz * self.n].cpu().numpy().view(np.int32).reshape(self.n, ssz // 4)
        return [dict(M=int(r[0]), status=int(r[1]), zmax=int(r[5]), hzmax=int(r[14]), env_size=int(r[15]), NR=int(r[18])) for r in state]

    def inspect(self):
        """Setup-phase outputs for parity tests: per candidate (N, M, status, pixel coordinates, grid points,
        CSR-like G~ rows), reconstructed PER PIXEL from the run-packed crop (a run = the region pixels of one image row inside one
        aligned 4-column cell; an entry of a run holds the weights of its four pixels for one grid point).  Reads the workspace back
        to the host."""
        lay = np.zeros(16, np.int64)
        _capi.check(_capi.lib().sdsm_plan_layout(self.plan, lay.ctypes.data_as(C.c_void_p)), 'sdsm_plan_layout')
        ws = self.ws.cpu().numpy()
        zcap, csz, nell, ssz, nruns = int(lay[9]), int(lay[11]), int(lay[13]), int(lay[14]), int(lay[15])
        cand = ws[lay[0]:lay[0] + csz * self.n].reshape(self.n, csz)
        i64 = lambda c, o: int(cand[c, o:o + 8].view(np.int64)[0])
        i32 = lambda c, o: int(cand[c, o:o + 4].view(np.int32)[0])
        state = ws[lay[1]:lay[1] + ssz * self.n].view(np.int32).reshape(self.n, ssz // 4)   # CandState
        crop_y = ws[lay[2]:lay[2] + 32 * nruns].view(np.float64).reshape(-1, 4)
        crop_rc = ws[lay[3]:lay[3] + 4 * nruns].view(np.uint32)
        run_meta = ws[lay[5]:lay[5] + 4 * nruns].view(np.uint32)
        xi_off = self.xi_offsets()
        grid = ws[lay[6]:].view(np.uint32)
        ell_im = ws[lay[7]:lay[7] + 4 * nell].view(np.uint32)
        ell_w = ws[lay[8]:lay[8] + 16 * nell].view(np.float32).reshape(-1, 4)
        out = []
        for i in range(self.n):
            N = int(self.n_pixels[i])
            M, status, hc, wc, npos = (int(v) for v in state[i, :5])
            NR = int(state[i, 18])
            ro, eo = i64(i, 96), i64(i, 8)
            g = grid[xi_off[i]:xi_off[i] + M]
            have_runs = status == 0 and NR > 0
            meta = run_meta[ro:ro + NR] if have_runs else np.zeros(0, np.uint32)
            rnnz, rhz, pm = (meta & 0xfff).astype(np.int64), ((meta >> 12) & 0xfff).astype(np.int64), (meta >> 24).astype(np.int64)
            rc = crop_rc[ro:ro + len(meta)]
            rows, cols, ys, nnzs, hnzs, idxs, wts = [], [], [], [], [], [], []
            zc = max(zcap, 1)
            for p in range(len(meta)):
                k = int(rnnz[p]) if M > 0 else 0
                im = ell_im[eo + np.arange(k) * NR + p] if k else np.zeros(0, np.uint32)
                w4 = ell_w[eo + np.arange(k) * NR + p] if k else np.zeros((0, 4), np.float32)
                for q in range(4):
                    if not (pm[p] >> q) & 1:
                        continue
                    rows.append(int(rc[p] >> 16)); cols.append(int(rc[p] & 0xffff) + q); ys.append(crop_y[ro + p, q])
                    sel = w4[:, q] != 0                                  # a grid point outside the pixel's window: weight 0
                    ci, cw = (im[sel] & 0xffff).astype(np.int64), w4[sel, q]
                    o2 = np.argsort(ci, kind='stable')                   # report rows by column index
                    col = np.zeros(zc, np.int64); wt = np.zeros(zc, np.float32)
                    col[:len(ci)] = ci[o2]; wt[:len(ci)] = cw[o2]
                    idxs.append(col); wts.append(wt); nnzs.append(len(ci))
                    hnzs.append(int((((im >> (16 + q)) & 1) != 0).sum()))
            rows, cols = np.asarray(rows, np.int64), np.asarray(cols, np.int64)
            o = np.lexsort((cols, rows))                                   # the crop is stored by runs in sorted scatter order; report it in raster order
            take = lambda a, dt: (np.asarray(a, dt)[o] if len(a) else np.zeros(0, dt))
            idx = np.stack(idxs, axis=1)[:, o] if idxs else np.zeros((zc, 0), np.int64)
            w = np.stack(wts, axis=1)[:, o] if wts else np.zeros((zc, 0), np.float32)
            out.append(dict(N=N, M=M, status=status, hc=hc, wc=wc, npos=npos, zmax=int(state[i, 5]), env_size=int(state[i, 15]), NR=NR,
                            y=take(ys, np.float64), r=rows[o] if len(rows) else rows, c=cols[o] if len(cols) else cols,
                            grid_r=(g >> 16).astype(np.int64), grid_c=(g & 0xffff).astype(np.int64), nnz=take(nnzs, np.int64), hnz=take(hnzs, np.int64),
                            run_nnz=rnnz.copy(), run_hnz=rhz.copy(), run_pixels=np.array([bin(int(v)).count('1') for v in pm], np.int64),
                            idx=idx.copy(), w=w.copy()))
        return out

    def fragments(self, records, select=None, masks=None, lazy=False):
        """Foreground fragments (bool arrays) and offsets, cropped from the bit-packed region-bbox masks."""
        if masks is None:
            masks = self.masks_dev.cpu().numpy()
        return fragments_from_masks(records, self.mask_info, self.mask_offset, masks, select, lazy)


def algorithmic_bytes(records, mask_info):
    """bytes_c = E_c (12 N_c + 8 (6 + M_c)) + 12 N_c + ceil(bbox_c / 8) + 128."""
    E = records['evals_value'].astype(np.int64) + records['evals_full'].astype(np.int64)
    N = records['n_pixels'].astype(np.int64)
    M = records['n_deform'].astype(np.int64)
    bbox = (mask_info[:len(records), 2].astype(np.int64) * mask_info[:len(records), 3].astype(np.int64) + 7) // 8
    return int((E * (12 * N + 8 * (6 + M)) + 12 * N + bbox + 128).sum())
