"""z-slab decomposition of the particle step over the GPUs of one node.

The reference (/root/reference/particle_solver) runs SPHSolver::SolveSPH (solver/sph_solver.cpp:87-98)
on one device.  Here the uniform grid is cut along z into slabs of whole cell layers, one process and
one solver per GPU.  A slab holds its own layers plus `halo` layers on each side that mirror the
neighbouring slabs; with halo = 2 the density of every particle within one cell of an owned particle is
computed locally from complete neighbourhoods, so one exchange per step is enough:

    after the integration each rank sends, to each z-neighbour, the particles it owned whose new cell
    layer lies inside the layers that neighbour holds (new owners and fresh halo copies alike), drops
    its old halo copies, and appends what it receives (fsa_slab_pack / fsa_slab_unpack), then runs the
    ordinary sort -> density -> force -> integration on its local set.

The records travel with torch.distributed point-to-point calls: RCCL over xGMI when the backend is
"nccl" (device buffers are handed to it directly), host staging when it is "gloo".  Because ties inside a
cell are broken by the order key each particle carries, a slab run gives, for every owned particle,
bit-for-bit the numbers of the single-device run.

Layer ownership is fixed for a run: the cuts balance the initial per-layer particle histogram.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

HALO = 2


# ------------------------------------------------------------------------------------------------
# layout
# ------------------------------------------------------------------------------------------------
def cell_layer_host(pos, vmin, vmax, dx, gridres):
    """Global cell layer (z index) of each position with the arithmetic of calcGridPos / calcHashD
    (solver/sph_kernel_impl.cuh:74-100): fp32 subtract, divide, floor; -1 outside the domain."""
    pos = np.asarray(pos)
    dt = pos.dtype.type
    inside = np.ones(len(pos), bool)
    for a in range(3):
        x = pos[:, a].astype(np.float64)
        inside &= (x < float(dt(vmax[a])) - 1e-10) & (x > float(dt(vmin[a])) + 1e-10)
    cell = [np.floor((pos[:, a] - dt(vmin[a])) / dt(dx)).astype(np.int64) for a in range(3)]
    for a in range(3):
        inside &= (cell[a] >= 0) & (cell[a] < gridres[a])
    return np.where(inside, cell[2], -1)


def owner_layer_host(pos, vmin, vmax, dx, gridres):
    """The cell layer a particle is filed under for ownership: its cell layer, or, outside the grid (where
    cell_layer_host gives -1), the layer of its z coordinate clamped into the grid.  Such a particle is nobody's
    neighbour, but its own sums run over the cells around it: it lives with the slab that owns this layer."""
    pos = np.asarray(pos)
    layer = cell_layer_host(pos, vmin, vmax, dx, gridres)
    dt = pos.dtype.type
    with np.errstate(invalid="ignore"):
        cz = np.floor((pos[:, 2] - dt(vmin[2])) / dt(dx))
    cz = np.clip(np.nan_to_num(cz, nan=0.0), 0, gridres[2] - 1).astype(np.int64)
    return np.where(layer >= 0, layer, cz)


def next_own(keep, left, right):
    """The layers a slab owns next, from the layers it and its two neighbours hold next (empty range: no
    neighbour): a neighbour's range reaches `halo` layers into the slab's own, its own `halo` layers beyond."""
    lo = (keep[0] + left[1]) // 2 if left[1] > left[0] else keep[0]
    hi = (right[0] + keep[1]) // 2 if right[1] > right[0] else keep[1]
    return lo, hi


def z_layer_host(z, vmin_z, vmax_z, dx, gz):
    """cell_layer_host for z coordinates alone (positions known to be inside the grid box in x and y)."""
    z = np.asarray(z, dtype=np.float32)
    f32 = np.float32
    z64 = z.astype(np.float64)
    inside = (z64 < float(f32(vmax_z)) - 1e-10) & (z64 > float(f32(vmin_z)) + 1e-10)
    cell = np.floor((z - f32(vmin_z)) / f32(dx)).astype(np.int64)
    inside &= (cell >= 0) & (cell < gz)
    return np.where(inside, cell, -1)


def balanced_cuts(layer_counts, nranks, min_layers=HALO + 1):
    """Cut indices c[0..nranks] (c[0] = 0, c[-1] = number of layers) that balance the particle counts,
    every slab at least `min_layers` thick."""
    gz = len(layer_counts)
    if gz < nranks * min_layers:
        raise ValueError(f"{gz} cell layers cannot be cut into {nranks} slabs of at least {min_layers}")
    pre = np.concatenate([[0], np.cumsum(layer_counts)]).astype(np.float64)
    total = pre[-1]
    cuts = [0]
    for r in range(1, nranks):
        target = total * r / nranks
        c = int(np.searchsorted(pre, target, side="left"))
        # the nearer of the two candidate layer boundaries
        if c > 0 and abs(pre[c - 1] - target) <= abs(pre[min(c, gz)] - target):
            c -= 1
        c = max(c, cuts[-1] + min_layers)
        c = min(c, gz - (nranks - r) * min_layers)
        cuts.append(c)
    cuts.append(gz)
    return cuts


class SlabLayout:
    def __init__(self, gz, cuts, halo=HALO):
        self.gz, self.cuts, self.halo = int(gz), [int(c) for c in cuts], int(halo)
        self.world = len(cuts) - 1

    def own(self, r):
        return (self.cuts[r], self.cuts[r + 1])

    def held(self, r):
        """layers a rank holds: its own plus the halo, clipped to the grid; empty for r outside the ranks"""
        if r < 0 or r >= self.world:
            return (0, 0)
        return (max(self.cuts[r] - self.halo, 0), min(self.cuts[r + 1] + self.halo, self.gz))


# ------------------------------------------------------------------------------------------------
# transport
# ------------------------------------------------------------------------------------------------
class NeighbourTransport:
    """Record exchange with the two z-neighbours over torch.distributed point-to-point calls."""

    def __init__(self, rank, world, device, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world, self.device, self.group = rank, world, device, group
        self.backend = dist.get_backend(group)
        # RCCL moves device memory itself; gloo needs host tensors
        self.staged = self.backend != "nccl" or device.type != "cuda"
        self.comm_device = torch.device("cpu") if self.staged else device

    def _batch(self, sends, recvs):
        dist = self.dist
        ops = [dist.P2POp(dist.irecv, t, peer, self.group) for peer, t in recvs]
        ops += [dist.P2POp(dist.isend, t, peer, self.group) for peer, t in sends]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    @staticmethod
    def _next_cap(count):
        """message size both ends agree on for the next step: this step's count plus head room"""
        return count + max(count // 8, 1024)

    def exchange(self, send, recv_bufs, record_bytes):
        """send = {peer: (uint8 tensor, count)}; recv_bufs = {peer: uint8 tensor}; returns {peer: count}.

        Every buffer starts with one header slot (a record's size) followed by the records, so the records
        received from `peer` are recv_bufs[peer][record_bytes : (1 + count) * record_bytes] when this
        returns (device buffers: stream-ordered on the current stream).

        The first exchange with a peer sends the count ahead in a message of its own.  After that a
        message has the size both ends derived from the previous count (`_next_cap`), its header slot
        carries the actual count, and only if that count does not fit is the rest sent in a second
        message: one round trip and one host synchronisation per step instead of two."""
        torch = self.torch
        rb = record_bytes
        peers = sorted(send)
        if not hasattr(self, "cap_out"):
            self.cap_out, self.cap_in = {}, {}
        counts = {}
        first = [p for p in peers if p not in self.cap_out]
        later = [p for p in peers if p in self.cap_out]

        def stage_out(t):
            return t.cpu() if self.staged else t

        def write_header(buf, count):
            buf[:8].copy_(torch.tensor([count], dtype=torch.int64).view(torch.uint8), non_blocking=True)

        def check_room(p, records):
            if (records + 1) * rb > recv_bufs[p].numel():
                raise RuntimeError(f"rank {self.rank}: {records} records from rank {p} exceed the receive buffer")

        # ---- peers seen for the first time: count, then payload --------------------------------
        if first:
            out_cnt = {p: torch.tensor([send[p][1]], dtype=torch.int64, device=self.comm_device) for p in first}
            in_cnt = {p: torch.zeros(1, dtype=torch.int64, device=self.comm_device) for p in first}
            self._batch([(p, out_cnt[p]) for p in first], [(p, in_cnt[p]) for p in first])
            for p in first:
                counts[p] = int(in_cnt[p].item())
                check_room(p, counts[p])
        # ---- one message per direction ----------------------------------------------------------
        sends, recvs, staged_in = [], [], {}
        for p in peers:
            buf, cnt = send[p]
            if p in first:
                n_out, n_in = cnt, counts[p]          # exact sizes, header unused
            else:
                n_out, n_in = self.cap_out[p], self.cap_in[p]
                write_header(buf, cnt)
                check_room(p, n_in)
            if p in later or n_out:
                sends.append((p, stage_out(buf[:(1 + n_out) * rb])))
            if p in later or n_in:
                if self.staged:
                    staged_in[p] = torch.empty((1 + n_in) * rb, dtype=torch.uint8)
                    recvs.append((p, staged_in[p]))
                else:
                    recvs.append((p, recv_bufs[p][:(1 + n_in) * rb]))
        self._batch(sends, recvs)
        # ---- headers, and the rare second message ------------------------------------------------
        sends, recvs, tails = [], [], {}
        for p in later:
            src = staged_in[p] if self.staged else recv_bufs[p]
            counts[p] = int(src[:8].view(torch.int64).item())
            check_room(p, counts[p])
            extra_in = counts[p] - self.cap_in[p]
            if extra_in > 0:
                if self.staged:
                    tails[p] = torch.empty(extra_in * rb, dtype=torch.uint8)
                    recvs.append((p, tails[p]))
                else:
                    lo = (1 + self.cap_in[p]) * rb
                    recvs.append((p, recv_bufs[p][lo:lo + extra_in * rb]))
            extra_out = send[p][1] - self.cap_out[p]
            if extra_out > 0:
                lo = (1 + self.cap_out[p]) * rb
                sends.append((p, stage_out(send[p][0][lo:lo + extra_out * rb])))
        self._batch(sends, recvs)
        for p, t in staged_in.items():
            keep = (1 + min(counts[p], self.cap_in.get(p, counts[p]))) * rb
            recv_bufs[p][:keep].copy_(t[:keep])
        for p, t in tails.items():
            lo = (1 + self.cap_in[p]) * rb
            recv_bufs[p][lo:lo + t.numel()].copy_(t)
        # never beyond the buffers (the two ends of a pair allocate theirs with the same size)
        for p in peers:
            self.cap_out[p] = min(self._next_cap(send[p][1]), send[p][0].numel() // rb - 1)
            self.cap_in[p] = min(self._next_cap(counts[p]), recv_bufs[p].numel() // rb - 1)
        return counts


    def redistribute(self, chunks, record_bytes):
        """chunks[d] = uint8 tensor of whole records for rank d (any rank, d != self.rank).  Returns
        {source rank: uint8 tensor} of what the other ranks had for this one.  Used when the slabs are
        re-cut; all ranks call it together."""
        torch, dist = self.torch, self.dist
        mine = torch.zeros(self.world, dtype=torch.int64, device=self.comm_device)
        for d, t in chunks.items():
            mine[d] = t.numel()
        table = [torch.zeros_like(mine) for _ in range(self.world)]
        dist.all_gather(table, mine, group=self.group)
        sends, recvs, got = [], [], {}
        for d, t in chunks.items():
            if d != self.rank and t.numel():
                sends.append((d, t.cpu() if self.staged else t.contiguous()))
        for src in range(self.world):
            nb = int(table[src][self.rank].item())
            if src != self.rank and nb:
                got[src] = torch.empty(nb, dtype=torch.uint8, device=self.comm_device)
                recvs.append((src, got[src]))
        self._batch(sends, recvs)
        return got

    def sum_over_ranks(self, values):
        """element-wise sum of an integer vector over all ranks, as a numpy array"""
        t = self.torch.as_tensor(np.asarray(values, dtype=np.int64)).to(self.comm_device)
        self.dist.all_reduce(t, group=self.group)
        return t.cpu().numpy()


# ------------------------------------------------------------------------------------------------
# local solver on the GPU
# ------------------------------------------------------------------------------------------------
class HipSlabBackend:
    """The slab's particles in one fsa_solver on a GPU; all work goes through the C ABI."""

    def __init__(self, params_dict, held, capacity, device_index, variant=None):
        import torch
        from . import capi
        self.torch, self.capi = torch, capi
        self.device = torch.device("cuda", device_index)
        torch.cuda.set_device(self.device)
        p = capi.make_params(capi.PRECISION_F32, **params_dict, slab_z0=held[0], slab_nz=held[1] - held[0])
        self.params = p
        self.solver = capi.Solver(p, capacity, device=device_index)
        # one torch stream carries the solver's kernels, the staging copies and (through torch's stream
        # semantics) the RCCL calls, so packing, transfer and unpacking are ordered without host syncs
        self.stream = torch.cuda.Stream(self.device)
        self.solver.set_stream(self.stream.cuda_stream)
        if variant is not None:
            self.solver.set_variant(variant)
        self.record_bytes = self.solver.slab_record_bytes

    def make_buffer(self, records):
        return self.torch.empty(max(records, 1) * self.record_bytes, dtype=self.torch.uint8, device=self.device)

    def stream_ctx(self):
        return self.torch.cuda.stream(self.stream)

    def upload(self, sub):
        self.solver.upload(sub)

    def set_scene_has_rigid(self, flag):
        self.solver.set_scene_has_rigid(flag)

    def set_owned(self, own):
        self.solver.slab_set_owned(own)

    def pack(self, own, keep, left, right, buf_left, cap_left, buf_right, cap_right):
        return self.solver.slab_pack(own, keep, left, right, buf_left.data_ptr(), cap_left, buf_right.data_ptr(),
                                     cap_right)

    def unpack(self, buf, count):
        self.solver.slab_unpack(buf.data_ptr(), count)

    def export_owned(self, own, buf, cap):
        return self.solver.slab_export_owned(own, buf.data_ptr(), cap)

    def step_local(self):
        self.solver.step(1)

    def synchronize(self):
        self.solver.synchronize()

    @property
    def n(self):
        return self.solver.n

    def get(self, name):
        return self.solver.get(name)

    def global_layer_of_sorted(self):
        """global cell layer of each local particle from the sorted hashes (-1: out of grid)"""
        h = self.solver.get("particleHash").astype(np.int64)
        gx, nz, z0 = self.params.gridres[0], self.params.slab_nz, self.params.slab_z0
        return np.where(h == self.capi.GRID_UNDEF, -1, (h // gx) % nz + z0)

    def close(self):
        self.solver.close()

    # -- the IISPH and DFSPH steps (IisphSlabs / DfsphSlabs below drive these) --------------------------
    real_bytes = 4
    carry_width = 1  # reals that travel beside a record: the pressure (IISPH) or v_star (DFSPH)

    def enable_iisph(self, group, normal):
        self.solver.slab_enable_iisph(group, normal)

    def enable_dfsph(self):
        self.solver.slab_enable_dfsph()
        self.carry_width = 3

    def pack_carry(self, own, keep, left, right, cap):
        """-> kept, lost, {-1: (record bytes, carried reals, count), +1: ...} for the sides that have a neighbour"""
        t = self.torch
        rb, w = self.record_bytes, self.carry_width
        sides = {-1: left[1] > left[0], 1: right[1] > right[0]}
        if not hasattr(self, "_send") or self._send_cap != (cap, w):
            self._send_cap = (cap, w)
            self._send = {sd: (t.empty(max(cap, 1) * rb, dtype=t.uint8, device=self.device),
                               t.empty(max(cap, 1) * w, dtype=t.float32, device=self.device)) for sd in (-1, 1)}
        (bl, xl), (br, xr) = self._send[-1], self._send[1]
        kept, nl, nr, lost = self.solver.slab_pack_carry(own, keep, left, right, bl.data_ptr(), xl.data_ptr(),
                                                         cap if sides[-1] else 0, br.data_ptr(), xr.data_ptr(),
                                                         cap if sides[1] else 0)
        out = {}
        if sides[-1]:
            out[-1] = (bl[:nl * rb], xl[:nl * w].view(t.uint8), nl)
        if sides[1]:
            out[1] = (br[:nr * rb], xr[:nr * w].view(t.uint8), nr)
        return kept, lost, out

    def unpack_carry(self, records, extras, count):
        if count:
            records, extras = records.to(self.device).contiguous(), extras.to(self.device).contiguous()
            self._hold = (records, extras)  # until the kernel has read them (same stream: the next call is ordered)
            self.solver.slab_unpack_carry(records.data_ptr(), extras.data_ptr(), count)

    def layer_list(self, lo, hi):
        """sorted positions of the in-grid particles of the global layers [lo, hi)"""
        t = self.torch
        cap = max(self.n, 1)
        idx = t.empty(cap, dtype=t.int32, device=self.device)
        cnt = self.solver.slab_layer_indices(lo, hi, idx.data_ptr(), cap) if hi > lo else 0
        return idx[:cnt]

    def gather(self, fields, idx):
        """the named fields of the listed particles, one after the other, as bytes"""
        t = self.torch
        cnt = idx.numel()
        parts = []
        for name in fields:
            buf = t.empty(cnt * IISPH_FIELD_WIDTH[name], dtype=t.float32, device=self.device)
            self.solver.slab_gather_field(name, idx.data_ptr(), cnt, buf.data_ptr())
            parts.append(buf.view(t.uint8))
        return t.cat(parts) if parts else t.empty(0, dtype=t.uint8, device=self.device)

    def scatter(self, fields, idx, data):
        t = self.torch
        cnt = idx.numel()
        data = data.to(self.device).contiguous()
        self._hold2 = data
        at = 0
        for name in fields:
            nb = cnt * IISPH_FIELD_WIDTH[name] * 4
            self.solver.slab_scatter_field(name, idx.data_ptr(), cnt, data.data_ptr() + at)
            at += nb
        assert at == data.numel(), "field refresh: the two slabs disagree on how many particles the layers hold"

    def field_bytes(self, fields, count):
        return count * sum(IISPH_FIELD_WIDTH[f] for f in fields) * 4

    def owned_error_sum(self, own):
        return self.solver.slab_owned_sum("error", own)

    def owned_error_max(self, own, start):
        return self.solver.slab_owned_max("error", own, start)

    def sort(self): self.solver.sort_particles()
    def setup_iisph(self): self.solver.setup_iisph()
    def iisph_factor(self): self.solver.iisph_factor()
    def iisph_npf(self): self.solver.non_pressure_force_multiphase()
    def iisph_predict_density(self): self.solver.iisph_predict_density()
    def iisph_calc_dijpjl(self): self.solver.iisph_calc_dijpjl()
    def iisph_calc_new_pressure(self): self.solver.iisph_calc_new_pressure()
    def iisph_update(self): self.solver.iisph_update()
    def dfsph_density_alpha(self): self.solver.compute_density_alpha()
    def dfsph_npf(self): self.solver.compute_non_pressure_force()
    def dfsph_solve_density_stiff(self): self.solver.solve_density_stiff()
    def dfsph_solve_divergence_stiff(self): self.solver.solve_divergence_stiff()
    def dfsph_apply_pstiff(self): self.solver.apply_pstiff()
    def dfsph_update_position(self): self.solver.update_position()
    def dfsph_update_velocities(self): self.solver.update_velocities()


IISPH_FIELD_WIDTH = {"density": 1, "pressure": 1, "restDensity": 1, "v_star": 3, "dii": 3, "dijpjl": 3, "pstiff": 1,
                     "DF_factor": 1}


# ------------------------------------------------------------------------------------------------
# the distributed solver
# ------------------------------------------------------------------------------------------------
class SlabSolver:
    """One rank's share of a scene.  `backend_factory(params_dict, held, capacity)` builds the local
    solver (HipSlabBackend on a GPU); `transport` moves records between ranks."""

    def __init__(self, scene, rank, world, transport, backend_factory, halo=HALO, slack=1.5, cuts=None,
                 exchange_buffers=True, layer_hist=None, has_rigid=None):
        """`scene` is the whole scene, or (with `layer_hist`, the particles per cell layer of the whole scene) at
        least the particles of the layers this rank owns, carrying their unique ids of the whole scene."""
        P = scene["params"]
        self.rank, self.world, self.transport = rank, world, transport
        pos = np.asarray(scene["pos"], dtype=np.float32)
        # grid geometry exactly as the solver derives it
        geo = derive_grid(P)
        self.gridres, self.dx = geo["gridres"], geo["dx"]
        gz = self.gridres[2]
        layer = cell_layer_host(pos, geo["vmin"], geo["vmax"], self.dx, self.gridres)
        hist = np.bincount(layer[layer >= 0], minlength=gz) if layer_hist is None else np.asarray(layer_hist)
        if cuts is None:
            cuts = balanced_cuts(hist, world, halo + 1) if world > 1 else [0, gz]
        self.layout = SlabLayout(gz, cuts, halo)
        own, held = self.layout.own(rank), self.layout.held(rank)
        self.own, self.held_layers = own, held
        # (out-of-grid particles go by the layer of their clamped z coordinate)
        filed = owner_layer_host(pos, geo["vmin"], geo["vmax"], self.dx, self.gridres)
        mine = (filed >= own[0]) & (filed < own[1])
        pre = np.concatenate([[0], np.cumsum(hist)])
        held_counts = [pre[self.layout.held(r)[1]] - pre[self.layout.held(r)[0]] for r in range(world)]
        n_outside = int((layer < 0).sum())
        self.capacity = int(max(held_counts) * slack) + n_outside + 4096
        widest = max(int(hist[max(c - 2 * halo, 0):c + 2 * halo].sum()) for c in self.layout.cuts[1:-1]) \
            if world > 1 else 0
        self.record_cap = int(widest * slack) + 4096
        self.params, self.factory, self.halo, self.slack = P, backend_factory, halo, slack
        self.geo = geo
        # (of the whole scene: a rank that was given its own particles only is told)
        self.has_rigid = bool((np.asarray(scene["type"]) == 4).any()) if has_rigid is None else bool(has_rigid)
        self.backend = backend_factory(P, held, self.capacity)
        uid = np.asarray(scene.get("uniqueId", np.arange(len(pos))), dtype=np.int32)
        sub = {k: np.ascontiguousarray(np.asarray(scene[k])[mine]) for k in ("pos", "vel", "type", "mass",
                                                                             "restDensity")}
        sub["uniqueId"] = np.ascontiguousarray(uid[mine])
        self.backend.upload(sub)
        self.backend.set_scene_has_rigid(self.has_rigid)
        self.backend.set_owned(own)
        self.peers = [p for p in (rank - 1, rank + 1) if 0 <= p < world]
        if exchange_buffers:  # not when a native slab group (capi.SlabGroup) moves the records
            self._make_exchange_buffers()
        self.lost = 0
        self.sent_records = 0
        self.recuts = 0

    def _make_exchange_buffers(self):
        # one header slot in front of the records (see NeighbourTransport.exchange)
        self.send_buf = {p: self.backend.make_buffer(self.record_cap + 1) for p in self.peers}
        self.recv_buf = {p: self.backend.make_buffer(self.record_cap + 1) for p in self.peers}
        self._null = self.backend.make_buffer(2)
        if self.transport is not None:
            self.transport.cap_out, self.transport.cap_in = {}, {}  # sizes are agreed afresh

    def _record_layers(self, rec):
        """global cell layer of each record (rows of a (count, record_bytes) uint8 tensor), -1 outside"""
        import torch
        pos = rec[:, :12].contiguous().view(torch.float32).view(-1, 3)
        f32 = torch.float32
        lo = torch.tensor([float(v) for v in self.geo["vmin"]], dtype=f32, device=pos.device)
        hi = torch.tensor([float(v) for v in self.geo["vmax"]], dtype=f32, device=pos.device)
        dx = torch.tensor(float(self.geo["dx"]), dtype=f32, device=pos.device)
        res = torch.tensor(self.gridres, dtype=torch.int64, device=pos.device)
        p64 = pos.double()
        inside = ((p64 < hi.double() - 1e-10) & (p64 > lo.double() + 1e-10)).all(1)
        cell = torch.floor((pos - lo) / dx).long()
        inside &= ((cell >= 0) & (cell < res)).all(1)
        self._record_filed = torch.where(inside, cell[:, 2], torch.nan_to_num(torch.floor((pos[:, 2] - lo[2]) / dx))
                                         .clamp(0, self.gridres[2] - 1).long())  # see owner_layer_host
        return torch.where(inside, cell[:, 2], torch.full_like(cell[:, 2], -1))

    def rebalance(self):
        """Re-cut the slabs so that every rank owns about the same number of particles again (all ranks
        call this together, between two steps).  Every particle goes to its new owner with its order
        key; the next exchange hands out the halo copies.  Returns True if the cuts changed."""
        import torch
        if self.world == 1:
            return False
        with self.backend.stream_ctx():
            b, rb = self.backend, self.backend.record_bytes
            cap = max(b.n, 1)
            buf = b.make_buffer(cap)
            cnt = b.export_owned(self.own, buf, cap)
            rec = buf[:cnt * rb].view(cnt, rb)
            layer = self._record_layers(rec)
            gz = self.gridres[2]
            local_hist = torch.bincount(layer[layer >= 0], minlength=gz).cpu().numpy()
            outside = int((layer < 0).sum().item())
            hist = self.transport.sum_over_ranks(np.concatenate([local_hist, [outside]]))
            hist, n_outside = hist[:gz], int(hist[gz])
            cuts = balanced_cuts(hist, self.world, self.halo + 1)
            if cuts == self.layout.cuts:
                return False
            bounds = torch.tensor(cuts[1:-1], dtype=torch.int64, device=layer.device)
            owner = torch.bucketize(self._record_filed, bounds, right=True)
            chunks = {d: rec[owner == d].reshape(-1) for d in range(self.world)}
            got = self.transport.redistribute({d: t for d, t in chunks.items() if d != self.rank}, rb)
            got[self.rank] = chunks[self.rank]
            # new layout, new local solver
            self.layout = SlabLayout(gz, cuts, self.halo)
            self.own, self.held_layers = self.layout.own(self.rank), self.layout.held(self.rank)
            pre = np.concatenate([[0], np.cumsum(hist)])
            held_counts = [pre[self.layout.held(r)[1]] - pre[self.layout.held(r)[0]] for r in range(self.world)]
            self.capacity = int(max(held_counts) * self.slack) + n_outside + 4096
            widest = max(int(hist[max(c - 2 * self.halo, 0):c + 2 * self.halo].sum()) for c in cuts[1:-1])
            self.record_cap = int(widest * self.slack) + 4096
            b.synchronize()
            b.close()
        self.backend = self.factory(self.params, self.held_layers, self.capacity)
        self.backend.set_scene_has_rigid(self.has_rigid)
        self.backend.set_owned(self.own)
        with self.backend.stream_ctx():
            for src in sorted(got):
                t = got[src]
                if t.numel():
                    dev = self.backend.make_buffer(1).device
                    t = t.to(dev).contiguous()
                    self.backend.unpack(t, t.numel() // rb)
            self.backend.synchronize()
        self._make_exchange_buffers()
        self.recuts += 1
        return True

    def exchange(self):
        """migration and halo refresh in one neighbour exchange"""
        r, lay = self.rank, self.layout
        left, right = lay.held(r - 1), lay.held(r + 1)
        bl = self.send_buf.get(r - 1, self._null)
        br = self.send_buf.get(r + 1, self._null)
        cl = self.record_cap if r - 1 in self.send_buf else 0
        cr = self.record_cap if r + 1 in self.send_buf else 0
        rb = self.backend.record_bytes
        kept, nl, nr, lost = self.backend.pack(self.own, self.held_layers, left, right, bl[rb:], cl, br[rb:], cr)
        self.lost += lost
        send = {}
        if r - 1 in self.send_buf:
            send[r - 1] = (bl, nl)
        if r + 1 in self.send_buf:
            send[r + 1] = (br, nr)
        if send:
            got = self.transport.exchange(send, self.recv_buf, self.backend.record_bytes)
            for p in self.peers:  # fixed order: left neighbour first
                self.backend.unpack(self.recv_buf[p][rb:], got[p])
        self.sent_records += nl + nr

    def step(self, nsteps=1):
        with self.backend.stream_ctx():
            for _ in range(nsteps):
                self.exchange()
                self.backend.step_local()

    def owned(self):
        """Results of the last step for the particles this rank owned during it, as host arrays."""
        b = self.backend
        layer = b.global_layer_of_sorted()
        m = ((layer >= self.own[0]) & (layer < self.own[1])) | (layer < 0)
        out = {k: b.get(k)[m] for k in ("pos", "vel", "density", "pressure", "force", "uniqueId", "type")}
        out["hash_layer"] = layer[m]
        return out

    def close(self):
        self.backend.close()


def rebalance_local(ranks):
    """SlabSolver.rebalance for slabs that all live in this process (a rehearsal of N ranks on one GPU, or one host
    program driving a node): every slab exports what it owns, the cuts are balanced afresh on the summed layer
    histogram, every slab gets a new solver for its new layers and the records that now belong to it.  Order keys
    travel along, so the run stays bit-identical.  Returns True if the cuts changed."""
    import torch
    world = len(ranks)
    if world == 1:
        return False
    recs, filed, hist, n_outside = [], [], 0, 0
    gz = ranks[0].gridres[2]
    for ss in ranks:
        with ss.backend.stream_ctx():
            b, rb = ss.backend, ss.backend.record_bytes
            cap = max(b.n, 1)
            buf = b.make_buffer(cap)
            cnt = b.export_owned(ss.own, buf, cap)
            rec = buf[:cnt * rb].view(cnt, rb)
            layer = ss._record_layers(rec)
            hist = hist + torch.bincount(layer[layer >= 0], minlength=gz).cpu().numpy()
            n_outside += int((layer < 0).sum().item())
            recs.append(rec)
            filed.append(ss._record_filed)
            b.synchronize()
    cuts = balanced_cuts(hist, world, ranks[0].halo + 1)
    if cuts == ranks[0].layout.cuts:
        return False
    bounds = torch.tensor(cuts[1:-1], dtype=torch.int64, device=filed[0].device)
    owners = [torch.bucketize(f, bounds, right=True) for f in filed]
    pre = np.concatenate([[0], np.cumsum(hist)])
    layout = SlabLayout(gz, cuts, ranks[0].halo)
    held_counts = [pre[layout.held(r)[1]] - pre[layout.held(r)[0]] for r in range(world)]
    capacity = int(max(held_counts) * ranks[0].slack) + n_outside + 4096
    widest = max(int(hist[max(c - 2 * layout.halo, 0):c + 2 * layout.halo].sum()) for c in cuts[1:-1])
    record_cap = int(widest * ranks[0].slack) + 4096
    for ss in ranks:
        ss.backend.close()
    for ss in ranks:
        ss.layout = layout
        ss.own, ss.held_layers = layout.own(ss.rank), layout.held(ss.rank)
        ss.capacity, ss.record_cap = capacity, record_cap
        ss.backend = ss.factory(ss.params, ss.held_layers, capacity)
        ss.backend.set_scene_has_rigid(ss.has_rigid)
        ss.backend.set_owned(ss.own)
        rb = ss.backend.record_bytes
        with ss.backend.stream_ctx():
            for src in range(world):  # in the order of the ranks the records come from
                t = recs[src][owners[src] == ss.rank].reshape(-1).contiguous()
                if t.numel():
                    ss.backend.unpack(t, t.numel() // rb)
            ss.backend.synchronize()
        ss.recuts += 1
    return True


# ------------------------------------------------------------------------------------------------
# the IISPH step on slabs
# ------------------------------------------------------------------------------------------------
class LocalComm:
    """Moves buffers between slabs that all live in this process."""

    def swap_counts(self, outgoing):
        return {(dst, src): n for (src, dst), n in outgoing.items()}

    def swap_bytes(self, outgoing, incoming_sizes):
        got = {(dst, src): t for (src, dst), t in outgoing.items()}
        for k, nb in incoming_sizes.items():
            assert k in got and got[k].numel() == nb, f"message {k[1]} -> {k[0]}: {got[k].numel()} bytes sent, {nb} expected"
        return got

    def sum(self, values):
        return float(sum(values))

    def max(self, values):
        return float(max(values))


class DistComm:
    """One slab per process: buffers travel between z-neighbours over torch.distributed point-to-point calls
    (RCCL for device tensors when the group's backend is "nccl", host staging over gloo otherwise)."""

    def __init__(self, transport):
        self.t = transport

    def swap_counts(self, outgoing):
        torch = self.t.torch
        dev = self.t.comm_device
        out = {k: torch.tensor([n], dtype=torch.int64, device=dev) for k, n in outgoing.items()}
        inc = {(src, dst): torch.zeros(1, dtype=torch.int64, device=dev) for (src, dst) in outgoing}
        self.t._batch([(dst, t) for (src, dst), t in out.items()], [(dst, t) for (src, dst), t in inc.items()])
        return {(src, dst): int(t.item()) for (src, dst), t in inc.items()}  # keyed (receiver, sender)

    def swap_bytes(self, outgoing, incoming_sizes):
        torch = self.t.torch
        dev = self.t.comm_device
        sends = [(dst, (t.cpu() if self.t.staged else t).contiguous()) for (src, dst), t in outgoing.items() if t.numel()]
        got = {k: torch.empty(nb, dtype=torch.uint8, device=dev) for k, nb in incoming_sizes.items()}
        recvs = [(src, t) for (dst, src), t in got.items() if t.numel()]
        self.t._batch(sends, recvs)
        return got

    def sum(self, values):
        torch = self.t.torch
        t = torch.tensor([float(sum(values))], dtype=torch.float64, device=self.t.comm_device)
        self.t.dist.all_reduce(t, group=self.t.group)
        return float(t.item())

    def max(self, values):
        torch = self.t.torch
        t = torch.tensor([float(max(values))], dtype=torch.float64, device=self.t.comm_device)
        self.t.dist.all_reduce(t, op=self.t.dist.ReduceOp.MAX, group=self.t.group)
        return float(t.item())


class _FieldSlabs:
    """What the IISPH and the DFSPH step on z-slabs share: the particle exchange with a field riding along, the lists
    of the layers next to the cuts, the refresh of named fields in the halo layers from their owners."""

    def __init__(self, members, comm):
        self.m, self.comm = list(members), comm
        self.lists = {}

    def _neighbours(self, ss):
        return [(sd, ss.rank + sd) for sd in (-1, 1) if 0 <= ss.rank + sd < ss.world]

    def exchange(self):
        counts, recs, extras, rb = {}, {}, {}, None
        for ss in self.m:
            with ss.backend.stream_ctx():
                lay = ss.layout
                kept, lost, out = ss.backend.pack_carry(ss.own, ss.held_layers, lay.held(ss.rank - 1),
                                                        lay.held(ss.rank + 1), ss.record_cap)
            ss.lost += lost
            rb = ss.backend.record_bytes
            for sd, peer in self._neighbours(ss):
                r, x, n = out[sd]
                counts[(ss.rank, peer)], recs[(ss.rank, peer)], extras[(ss.rank, peer)] = n, r, x
                ss.sent_records += n
        for ss in self.m:
            ss.backend.synchronize()
        inc = self.comm.swap_counts(counts)
        got_r = self.comm.swap_bytes(recs, {k: n * rb for k, n in inc.items()})
        got_x = self.comm.swap_bytes(extras, {k: n * self.m[0].backend.real_bytes * self.m[0].backend.carry_width
                                               for k, n in inc.items()})
        for ss in self.m:
            with ss.backend.stream_ctx():
                for sd, peer in self._neighbours(ss):  # fixed order: left neighbour first
                    k = (ss.rank, peer)
                    ss.backend.unpack_carry(got_r[k], got_x[k], inc[k])

    def _after_sort(self):
        """the lists of the layers next to the cuts: what a slab owns there, and its copies of what the neighbour owns"""
        edge_counts = {}
        for ss in self.m:
            h = ss.halo
            with ss.backend.stream_ctx():
                for sd, peer in self._neighbours(ss):
                    if sd < 0:
                        edge, halo = (ss.own[0], min(ss.own[0] + h, ss.own[1])), (ss.own[0] - h, ss.own[0])
                    else:
                        edge, halo = (max(ss.own[1] - h, ss.own[0]), ss.own[1]), (ss.own[1], ss.own[1] + h)
                    self.lists[(ss.rank, sd, "edge")] = ss.backend.layer_list(*edge)
                    self.lists[(ss.rank, sd, "halo")] = ss.backend.layer_list(*halo)
                    edge_counts[(ss.rank, peer)] = int(self.lists[(ss.rank, sd, "edge")].shape[0])
        inc = self.comm.swap_counts(edge_counts)
        for ss in self.m:
            for sd, peer in self._neighbours(ss):
                mine = int(self.lists[(ss.rank, sd, "halo")].shape[0])
                if inc[(ss.rank, peer)] != mine:
                    raise RuntimeError(f"slab {ss.rank}: {mine} halo copies of the layers slab {peer} owns next to the cut, "
                                       f"but that slab holds {inc[(ss.rank, peer)]} particles there")

    def refresh(self, fields):
        out, sizes = {}, {}
        for ss in self.m:
            with ss.backend.stream_ctx():
                for sd, peer in self._neighbours(ss):
                    out[(ss.rank, peer)] = ss.backend.gather(fields, self.lists[(ss.rank, sd, "edge")])
                    sizes[(ss.rank, peer)] = ss.backend.field_bytes(fields, int(self.lists[(ss.rank, sd, "halo")].shape[0]))
        for ss in self.m:
            ss.backend.synchronize()
        got = self.comm.swap_bytes(out, sizes)
        for ss in self.m:
            with ss.backend.stream_ctx():
                for sd, peer in self._neighbours(ss):
                    ss.backend.scatter(fields, self.lists[(ss.rank, sd, "halo")], got[(ss.rank, peer)])

    def _each(self, name):
        for ss in self.m:
            with ss.backend.stream_ctx():
                getattr(ss.backend, name)()


class IisphSlabs(_FieldSlabs):
    """MultiphaseSPHSolver::SolveIISPH (solver/MultiphaseSPHSolver.cpp:173-188; rounds: solver/sph_solver.cu:1409-1443)
    on z-slabs: the slabs of `members` (SlabSolver objects built with iisph=True: all the slabs of the scene with a
    LocalComm, or this process's one with a DistComm).

    A step runs some 25 neighbour passes between two sorts, each reading what the pass before wrote at the
    neighbours, so after every pass the fields it wrote are refreshed in the halo layers from the slab that owns
    the particles: the sorted order of a slab is (y, layer, x, order key), so both slabs hold the particles of the
    layers next to a cut in the same order, and a refresh is a gather on one side and a scatter on the other.  The
    particle exchange after the integration is the WCSPH one with the pressure riding along (the relaxation of a step
    starts from the pressures of the step before).  The mean of error[] that ends the pressure rounds is summed over
    the particles each slab owns and then over the slabs.  Particle counts are known on the host here (one wait per
    exchange and per round): the step is a dozen times longer than the WCSPH one."""

    def __init__(self, members, comm, n_global, group, normal):
        super().__init__(members, comm)
        self.n_global = int(n_global)
        for ss in self.m:
            ss.backend.enable_iisph(group, normal)
        self.rounds = []

    def setup(self):
        """SetupIISPH (solver/MultiphaseSPHSolver.cpp:850-860): sort, factor, rigid particle volumes"""
        self.exchange()  # hands out the halo copies
        self._each("setup_iisph")
        self._after_sort()
        self.refresh(["density", "dii", "restDensity"])

    def step(self, nsteps=1):
        for _ in range(nsteps):
            self._each("iisph_npf")
            self.refresh(["v_star"])
            self._each("iisph_predict_density")
            last, rounds = np.float32(0.0), 0
            for _it in range(10):  # "temp 10", solver/sph_solver.cu:1416
                self._each("iisph_calc_dijpjl")
                self.refresh(["dijpjl"])
                self._each("iisph_calc_new_pressure")
                self.refresh(["pressure"])
                total = self.comm.sum([ss.backend.owned_error_sum(ss.own) for ss in self.m])
                mean = np.float32(total / self.n_global)
                rounds += 1
                if abs(float(mean)) < 1e-6 or abs(float(np.float32(mean - last))) < 1e-6:  # :1440
                    break
                last = mean
            self.rounds.append(rounds)
            self._each("iisph_update")
            self.exchange()
            self._each("sort")
            self._each("iisph_factor")
            self._after_sort()
            self.refresh(["density", "dii"])


class DfsphSlabs(_FieldSlabs):
    """SPHSolver::SolveDFSPH (solver/sph_solver.cpp:108-120; the two correction loops: solver/sph_solver.cu:144-256) on
    z-slabs, with the pattern of IisphSlabs: v_star is refreshed in the halo layers after computeNonPForce and after
    every applyPStiff, pstiff after every solveDensityStiff / solveDivergenceStiff, density and DF_factor after
    computeDensityAlpha; v_star rides with the exchange records (the divergence loop after the sort goes on from it);
    a loop ends when the maximum of error[] over the owned particles of all slabs is below the threshold."""

    def __init__(self, members, comm, maxiter=5, ethres=1.0):
        super().__init__(members, comm)
        self.maxiter, self.ethres = int(maxiter), float(ethres)
        for ss in self.m:
            ss.backend.enable_dfsph()
        self.density_rounds, self.divergence_rounds = [], []

    def _sorted(self):
        self._each("sort")
        self._each("dfsph_density_alpha")
        self._after_sort()
        self.refresh(["density", "DF_factor"])

    def setup(self):
        """SetupDFSPH (solver/sph_solver.cpp:102-106): sort + computeDensityAlpha"""
        self.exchange()  # hands out the halo copies
        self._sorted()

    def _loop(self, solve, start):
        applied = 0
        for _it in range(self.maxiter):
            self._each(solve)
            self.refresh(["pstiff"])
            worst = self.comm.max([ss.backend.owned_error_max(ss.own, start) for ss in self.m])
            if worst < self.ethres:  # solver/sph_solver.cu:176-180
                break
            self._each("dfsph_apply_pstiff")
            self.refresh(["v_star"])
            applied += 1
        return applied

    def step(self, nsteps=1):
        for _ in range(nsteps):
            self._each("dfsph_npf")
            self.refresh(["v_star"])
            self.density_rounds.append(self._loop("dfsph_solve_density_stiff", -9999.0))
            self._each("dfsph_update_position")
            self.exchange()
            self._sorted()
            self._each("dfsph_solve_divergence_stiff")  # the extra first launch of solver/sph_solver.cu:226
            self.refresh(["pstiff"])
            self.divergence_rounds.append(self._loop("dfsph_solve_divergence_stiff", 0.0))
            self._each("dfsph_update_velocities")


def derive_grid(P):
    """Grid box, cell size and resolution as MultiphaseSPHSolver::LoadParam derives them in fp32
    (solver/MultiphaseSPHSolver.cpp:381, 452-466); computed by the library's fsa_derive_params."""
    from . import capi
    p = capi.make_params(capi.PRECISION_F32, **P)
    return {"dx": np.float32(p.dx), "vmin": [np.float32(v) for v in p.gridxmin],
            "vmax": [np.float32(v) for v in p.gridxmax], "gridres": [int(v) for v in p.gridres]}


# ------------------------------------------------------------------------------------------------
# bench entry (bench.py --gpus N under torch.distributed.run)
# ------------------------------------------------------------------------------------------------
def bench_distributed(args, build_scene):
    """One rank of `bench.py --gpus N`.  The communication libraries print connection chatter to file
    descriptor 1; it is pointed at stderr for the duration, so that stdout carries the bench line only."""
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        return _bench_distributed(args, build_scene)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        os.close(saved_stdout)


class _Watchdog:
    """Ends the process (status 124) when a phase of the multi-rank bench makes no progress for `limit` seconds: a send
    without its receive would otherwise leave every rank waiting inside the library for good.  A thread, because a
    signal handler does not run while the main thread waits inside a library call."""

    def __init__(self, rank, limit):
        import threading
        self.rank, self.limit, self.phase, self.deadline = rank, limit, "start", time.monotonic() + limit
        self.done = threading.Event()
        if limit > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def kick(self, phase):
        self.phase, self.deadline = phase, time.monotonic() + self.limit

    def stop(self):
        self.done.set()

    def _run(self):
        while not self.done.wait(1.0):
            if time.monotonic() > self.deadline:
                print(f"rank {self.rank}: no progress for {self.limit:.0f} s in phase '{self.phase}' of the slab bench; "
                      "giving up", file=sys.stderr, flush=True)
                os._exit(124)


def _bench_distributed(args, build_scene):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    dog = _Watchdog(rank, float(os.environ.get("FSA_BENCH_WATCHDOG", "600")))
    ndev = torch.cuda.device_count()
    assert ndev > 0, "bench.py needs a GPU"
    device_index = local % ndev
    torch.cuda.set_device(device_index)
    device = torch.device("cuda", device_index)
    # Control traffic (barriers, statistics) goes over a gloo group; the records go over RCCL whenever
    # every rank has a GPU of its own, whatever the launcher did to device visibility.  Ranks sharing a
    # card (rehearsal on a one-GPU box) stage the records through the host over gloo instead.
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # a rank that died must not leave the others waiting for the default half hour
        import datetime
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))
    props = torch.cuda.get_device_properties(device_index)
    ident = "|".join(str(getattr(props, a, "")) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    if not ident.strip("|"):
        ident = f"visible-index-{device_index}-of-{ndev}"
    idents = [None] * world
    dist.all_gather_object(idents, ident)
    forced = os.environ.get("FSA_SLAB_BACKEND")  # "nccl" or "gloo" overrides the detection
    use_nccl = (len(set(idents)) == world) if forced is None else forced == "nccl"
    record_group = None
    if use_nccl:
        # The communicator is created here and now (device_id given).  A rank with a GPU of its own that cannot make
        # it is an error, not a reason to measure something else: the job ends non-zero on every rank.  (The host-staged
        # transport is for ranks that share a card, and for FSA_SLAB_BACKEND=gloo.)
        try:
            record_group = dist.new_group(backend="nccl", device_id=device)
            ok = 1
        except Exception as exc:  # noqa: BLE001
            print(f"rank {rank}: RCCL group creation failed ({exc})", file=sys.stderr)
            ok = 0
        oks = [None] * world
        dist.all_gather_object(oks, ok)
        if not all(oks):
            raise SystemExit(f"rank {rank}: ranks {[r for r, o in enumerate(oks) if not o]} could not make the RCCL "
                             "communicator; set FSA_SLAB_BACKEND=gloo to stage the records through the host instead")
    backend = "nccl" if use_nccl else "gloo"
    # Every rank builds the particles of its own layers only: first the scene's per-layer histogram (the fluid
    # lattice contributes one count per z value, the walls are a thin shell), from it the cuts, then the window.
    # (A jittered scene draws its noise for the whole scene in one stream: built whole.)
    layer_hist, cuts0 = None, None
    if args.jitter > 0:
        sc = build_scene(args.scene, args.jitter)
        n = len(sc["pos"])
    else:
        probe = build_scene(args.scene, 0.0, z_select=lambda z: np.zeros(len(z), bool))
        geo = derive_grid(probe["params"])
        gz = geo["gridres"][2]
        layer_of = lambda z: z_layer_host(z, geo["vmin"][2], geo["vmax"][2], geo["dx"], gz)  # noqa: E731
        zv, zc = probe["z_counts"]
        lay = layer_of(zv)
        assert (lay >= 0).all(), "the generated scenes lie inside the grid box"
        layer_hist = np.bincount(lay, weights=zc, minlength=gz).astype(np.int64)
        cuts0 = balanced_cuts(layer_hist, world, HALO + 1) if world > 1 else [0, gz]
        lo, hi = cuts0[rank], cuts0[rank + 1]
        sc = build_scene(args.scene, 0.0, z_select=lambda z: (layer_of(z) >= lo) & (layer_of(z) < hi))
        n = sc["n_total"]
    n_fluid = sc["n_fluid"]
    dog.kick("solver and group set-up")
    transport = NeighbourTransport(rank, world, device, group=record_group)
    # With a GPU per rank the step runs in the library's slab group: pack, ncclSend / ncclRecv, unpack and the
    # solver's step queued without a host round trip.  Ranks sharing a card (rehearsal) keep the torch transport.
    rigid = torch.tensor([int((np.asarray(sc["type"]) == 4).any())])
    dist.all_reduce(rigid, op=dist.ReduceOp.MAX)  # the ranks hold different parts of the scene and must agree
    has_rigid = bool(rigid.item())
    # (ranks that share a card, or FSA_SLAB_BACKEND=gloo: the same group of the library with its messages staged through
    # the host and this job's gloo group: one host wait per message, for rehearsals; FSA_SLAB_NATIVE=0: the Python driver)
    native = os.environ.get("FSA_SLAB_NATIVE", "1") != "0" and not has_rigid and (args.variant is None or args.variant == 3)
    hosted = native and not use_nccl
    ss = SlabSolver(sc, rank, world, transport,
                    lambda P, held, cap: HipSlabBackend(P, held, cap, device_index, args.variant),
                    exchange_buffers=not native, cuts=cuts0, layer_hist=layer_hist, has_rigid=has_rigid)
    del sc
    sync_dev = torch.device("cpu")
    from . import capi

    def make_group():
        if hosted:
            return capi.SlabGroup.hosted(ss.backend.solver, rank, world, capi.torch_host_transport(), ss.layout.cuts, ss.halo,
                                         ss.record_cap)
        uid = [None]
        if rank == 0:
            try:
                uid = [capi.SlabGroup.unique_id()]
            except Exception as exc:  # noqa: BLE001  (the other ranks wait in the broadcast: tell them)
                print(f"rank 0: no RCCL unique id ({exc})", file=sys.stderr)
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is None:
            raise RuntimeError("rank 0 could not make an RCCL unique id")
        return capi.SlabGroup.rccl(ss.backend.solver, rank, world, uid[0], ss.layout.cuts, ss.halo, ss.record_cap)

    group = None
    if native:
        # every rank has its own GPU here; a rank that cannot make the library's group ends the job non-zero on all
        # ranks (FSA_SLAB_NATIVE=0 selects the torch transport on purpose)
        err = None
        try:
            group = make_group()
        except Exception as exc:  # noqa: BLE001
            err = str(exc)
            print(f"rank {rank}: the library's slab group could not be made ({exc})", file=sys.stderr)
        made = [None] * world
        dist.all_gather_object(made, group is not None)
        if not all(made):
            if group is not None:
                group.close()
            raise SystemExit(f"rank {rank}: the library's slab group could not be made on ranks "
                             f"{[r for r, m in enumerate(made) if not m]}" + (f" ({err})" if err else ""))
    recut_every = getattr(args, "recut_every", 0) or 0

    def run(nsteps):
        nonlocal group
        done = 0
        while done < nsteps:
            k = nsteps - done
            if recut_every:
                k = min(k, recut_every - (run.total % recut_every))
            if native:
                group.step(k)
            else:
                ss.step(k)
            done += k
            run.total += k
            if recut_every and run.total % recut_every == 0:
                if native:
                    st = group.sync()
                    run.lost += int(st.lost)
                    run.sent += int(st.records_sent)
                    group.close()
                timing = ss.backend.solver.timing_read() if run.timing else None
                ss.rebalance()
                if run.timing:  # the re-cut builds a new solver: carry the kernel times over
                    for name, (c, ms) in timing.items():
                        c0, m0 = run.kernels.get(name, (0, 0.0))
                        run.kernels[name] = (c0 + c, m0 + ms)
                    ss.backend.solver.timing_enable(True)
                    ss.backend.solver.timing_reset()
                if native:
                    group = make_group()
    run.total, run.lost, run.sent, run.timing, run.kernels = 0, 0, 0, False, {}

    def barrier():
        if native:
            group.sync()
        torch.cuda.synchronize()
        t = torch.zeros(1, device=sync_dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()

    dog.kick("spin-up")
    run(args.spinup)
    dog.kick("warm-up")
    run(args.warmup)
    barrier()
    dog.kick("timed steps")
    ss.backend.solver.timing_enable(True)
    ss.backend.solver.timing_reset()
    run.timing = True
    recuts0, sent0 = ss.recuts, ss.sent_records
    if native:
        run.sent = -int(group.sync().records_sent)
    barrier()
    host_waits0 = capi.load().fsa_debug_host_waits()
    t0 = time.perf_counter()
    run(args.steps)
    host_waits = capi.load().fsa_debug_host_waits() - host_waits0
    barrier()
    wall = time.perf_counter() - t0
    dog.kick("statistics")
    kt = ss.backend.solver.timing_read()
    for name, (c, ms) in run.kernels.items():
        c0, m0 = kt.get(name, (0, 0.0))
        kt[name] = (c0 + c, m0 + ms)
    ss.backend.solver.timing_enable(False)
    if native:
        st = group.sync()
        lost = run.lost + int(st.lost)
        sent = run.sent + int(st.records_sent)
    else:
        lost, sent = ss.lost, ss.sent_records - sent0
    local_n = ss.backend.n
    types = ss.backend.get("type")
    local_fluid = int((types == 0).sum())
    layer = ss.backend.global_layer_of_sorted()
    owned_n = int((((layer >= ss.own[0]) & (layer < ss.own[1])) | (layer < 0)).sum())
    exch_ms = sum(ms for k, (c, ms) in kt.items() if k in ("slab_pack", "edge_send", "exchange", "slab_unpack", "unpack_hash")) / max(args.steps, 1)
    stats = torch.tensor([wall, float(local_n), float(lost), float(ss.backend.solver.num_undef()), float(sent),
                          exch_ms, float(host_waits), float(owned_n)], dtype=torch.float64, device=sync_dev)
    mx = stats.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sm = stats.clone()
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    res = dict(n=n, n_fluid=n_fluid, wall=float(mx[0]), kernels=kt, num_cells=ss.backend.params.num_cells,
               n_undef=int(sm[3]), gridres=list(ss.gridres), local_n=local_n, local_n_fluid=local_fluid,
               slab_cuts=ss.layout.cuts, halo_layers=ss.layout.halo, max_local_particles=int(mx[1]),
               mean_local_particles=float(sm[1]) / world, sum_local_particles=int(sm[1]), lost_particles=int(sm[2]),
               records_sent_per_step=float(sm[4]) / max(args.steps, 1), comm_backend=backend,
               slab_step=("library slab group, messages staged through the host over gloo (one host wait per message)"
                          if hosted else "library slab group (RCCL send/recv, no host round trip)") if native else
                         "torch.distributed transport, two host round trips per step",
               exchange_ms_per_step_max=float(mx[5]), host_syncs_in_timed_steps=int(mx[6]),
               recuts=ss.recuts - recuts0, recut_every=recut_every, sum_owned_particles=int(sm[7]))
    if native:
        group.close()
    ss.close()
    dist.barrier()
    dist.destroy_process_group()
    dog.stop()
    return res
