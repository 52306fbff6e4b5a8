"""CPU stand-in for the per-GPU solver of fluidsolver_amd.slabs, built on the oracle.

Test infrastructure: lets the slab driver (layout, neighbour exchange, ownership rules) run under gloo
without a GPU.  It restates fsa_slab_pack / fsa_slab_unpack and the order-key tie break in numpy.
"""
import contextlib

import numpy as np
import torch

GRID_UNDEF = 99999999
RECORD = np.dtype([("posType", "<f4", 4), ("velMass", "<f4", 4), ("orderKey", "<i8"), ("restDensity", "<f4"),
                   ("uniqueId", "<i4")])
assert RECORD.itemsize == 48


class OracleSlabBackend:
    record_bytes = RECORD.itemsize

    def __init__(self, orc, params_dict, held, capacity):
        self.orc = orc
        P = params_dict
        self.p = orc.make_params(P["spacing"], P["smoothratio"], P["vmin"], P["vmax"], gravity=P["gravity"],
                                 dt=P["dt"], viscosity=P["viscosity"], pressureK=P["pressureK"],
                                 boundary_visc=P["boundary_visc"], acceleration_limit=P["acceleration_limit"],
                                 boundary_mode=P["boundary_mode"])
        self.gres = [int(v) for v in self.p.gridres]
        self.num_cells = self.gres[0] * self.gres[1] * self.gres[2]
        self.held, self.capacity = held, capacity
        self.a = {"pos": np.zeros((0, 3), np.float32), "vel": np.zeros((0, 3), np.float32),
                  "type": np.zeros(0, np.int32), "mass": np.zeros(0, np.float32),
                  "restDensity": np.zeros(0, np.float32), "uniqueId": np.zeros(0, np.int32)}
        self.key = np.zeros(0, np.int64)
        self.sorted_hash = None
        self.res = {}

    # -- plumbing ---------------------------------------------------------------------------------
    def make_buffer(self, records):
        return torch.empty(max(records, 1) * self.record_bytes, dtype=torch.uint8)

    def stream_ctx(self):
        return contextlib.nullcontext()

    def set_scene_has_rigid(self, flag):
        pass

    def set_owned(self, own):
        pass

    def synchronize(self):
        pass

    def close(self):
        pass

    @property
    def n(self):
        return len(self.key)

    def upload(self, sub):
        self.a = {"pos": np.array(sub["pos"], np.float32).reshape(-1, 3),
                  "vel": np.array(sub["vel"], np.float32).reshape(-1, 3),
                  "type": np.array(sub["type"], np.int32), "mass": np.array(sub["mass"], np.float32),
                  "restDensity": np.array(sub["restDensity"], np.float32),
                  "uniqueId": np.array(sub["uniqueId"], np.int32)}
        self.key = self.a["uniqueId"].astype(np.int64)
        self.sorted_hash = None

    def _state(self):
        # the oracle indexes indexTable with uniqueId, so it gets local ids; the global ones stay here
        st = self.orc.new_state(self.n, self.num_cells)
        st.load({k: self.a[k] for k in ("pos", "vel", "type", "mass", "restDensity")})
        return st

    def _layer_of_hash(self, h):
        h = h.astype(np.int64)
        return np.where(h == GRID_UNDEF, -1, (h // self.gres[0]) % self.gres[2])

    # -- exchange ---------------------------------------------------------------------------------
    def _classify(self, own, keep, left, right):
        """masks (stays, goes left, goes right, lost) over the present arrays: the rule of fsa_slab_pack"""
        n = self.n
        if self.sorted_hash is None:
            owned = np.ones(n, bool)
        else:
            was = self._layer_of_hash(self.sorted_hash)
            owned = (was < 0) | ((was >= own[0]) & (was < own[1]))
        st = self._state()
        st.calc_hash(self.p)
        now = self._layer_of_hash(st["particleHash"].copy())
        st.close()
        inside = now >= 0
        # an out-of-grid particle lives with the slab that owns the layer of its clamped z coordinate
        from fluidsolver_amd.slabs import next_own
        with np.errstate(invalid="ignore"):
            cz = np.floor((self.a["pos"][:, 2] - np.float32(self.p.gridxmin[2])) / np.float32(self.p.dx))
        filed = np.clip(np.nan_to_num(cz, nan=0.0), 0, self.gres[2] - 1).astype(np.int64)
        lo, hi = next_own(keep, left, right)
        k = owned & np.where(inside, (now >= keep[0]) & (now < keep[1]), (filed >= lo) & (filed < hi))
        l = owned & np.where(inside, (now >= left[0]) & (now < left[1]), filed < lo)
        r = owned & np.where(inside, (now >= right[0]) & (now < right[1]), filed >= hi)
        lost = owned & inside & ~k & ~l & ~r
        return k, l, r, lost

    def _records(self, mask):
        rec = np.zeros(int(mask.sum()), RECORD)
        rec["posType"][:, :3] = self.a["pos"][mask]
        rec["posType"][:, 3] = self.a["type"][mask].view(np.float32)
        rec["velMass"][:, :3] = self.a["vel"][mask]
        rec["velMass"][:, 3] = self.a["mass"][mask]
        rec["orderKey"] = self.key[mask]
        rec["restDensity"] = self.a["restDensity"][mask]
        rec["uniqueId"] = self.a["uniqueId"][mask]
        return rec

    def pack(self, own, keep, left, right, buf_left, cap_left, buf_right, cap_right):
        k, l, r, lost = self._classify(own, keep, left, right)
        for mask, buf, cap in ((l, buf_left, cap_left), (r, buf_right, cap_right)):
            m = int(mask.sum())
            assert m <= cap, "send buffer too small"
            if m:
                buf.numpy()[:m * RECORD.itemsize] = self._records(mask).view(np.uint8).reshape(-1)
        self.a = {name: v[k] for name, v in self.a.items()}
        self.key = self.key[k]
        self.sorted_hash = None
        return int(k.sum()), int(l.sum()), int(r.sum()), int(lost.sum())

    def export_owned(self, own, buf, cap):
        n = self.n
        if self.sorted_hash is None:
            owned = np.ones(n, bool)
        else:
            was = self._layer_of_hash(self.sorted_hash)
            owned = (was < 0) | ((was >= own[0]) & (was < own[1]))
        m = int(owned.sum())
        assert m <= cap
        rec = np.zeros(m, RECORD)
        rec["posType"][:, :3] = self.a["pos"][owned]
        rec["posType"][:, 3] = self.a["type"][owned].view(np.float32)
        rec["velMass"][:, :3] = self.a["vel"][owned]
        rec["velMass"][:, 3] = self.a["mass"][owned]
        rec["orderKey"] = self.key[owned]
        rec["restDensity"] = self.a["restDensity"][owned]
        rec["uniqueId"] = self.a["uniqueId"][owned]
        if m:
            buf.numpy()[:m * RECORD.itemsize] = rec.view(np.uint8).reshape(-1)
        return m

    def unpack(self, buf, count):
        if count == 0:
            return
        rec = buf.numpy()[:count * RECORD.itemsize].view(RECORD)
        add = {"pos": rec["posType"][:, :3], "type": np.ascontiguousarray(rec["posType"][:, 3]).view(np.int32),
               "vel": rec["velMass"][:, :3], "mass": rec["velMass"][:, 3], "restDensity": rec["restDensity"],
               "uniqueId": rec["uniqueId"]}
        self.a = {name: np.concatenate([v, add[name]]) for name, v in self.a.items()}
        self.key = np.concatenate([self.key, rec["orderKey"]])

    # -- one step ---------------------------------------------------------------------------------
    def step_local(self):
        # the oracle's sort is stable in array order: put the array in key order first
        order = np.argsort(self.key.astype(np.uint64), kind="stable")
        self.a = {name: v[order] for name, v in self.a.items()}
        st = self._state()
        st.sort_particles(self.p)
        h = st["particleHash"].copy()
        st.compute_pressure(self.p)
        st.compute_force(self.p)
        self.res = {k: st[k].copy() for k in ("density", "pressure", "force")}
        st.advect(self.p)
        local = st["uniqueId"].copy()
        for k in ("pos", "vel", "type", "mass", "restDensity"):
            self.a[k] = st[k].copy()
        self.a["uniqueId"] = self.a["uniqueId"][local]
        st.close()
        # new keys: (global cell, rank inside the cell); the oracle holds the whole grid, so h is global
        n = len(h)
        first = np.ones(n, bool)
        first[1:] = h[1:] != h[:-1]
        start = np.maximum.accumulate(np.where(first, np.arange(n), 0))
        self.key = (h.astype(np.int64) << 32) | (np.arange(n) - start)
        self.sorted_hash = h

    def get(self, name):
        if name in self.res:
            return self.res[name]
        if name == "particleHash":
            return self.sorted_hash
        return self.a[name]

    def global_layer_of_sorted(self):
        return self._layer_of_hash(self.sorted_hash)


IISPH_FIELD_WIDTH = {"density": 1, "pressure": 1, "restDensity": 1, "v_star": 3, "dii": 3, "dijpjl": 3, "pstiff": 1,
                     "DF_factor": 1}


class OracleIisphSlabBackend(OracleSlabBackend):
    """the same stand-in for the IISPH step on slabs (fluidsolver_amd.slabs.IisphSlabs): between a sort and the next
    exchange the slab's particles live in one oracle state, on which the driver runs the oracle's stage functions;
    the field refresh and the pressure that rides along with the records are restated in numpy"""
    real_bytes = 4
    carry_width = 1

    def __init__(self, orc, params_dict, held, capacity):
        super().__init__(orc, params_dict, held, capacity)
        P = params_dict
        self.p.surface_tension = P.get("surface_tension", 0.0)
        self.p.boundary_friction = P.get("boundary_friction", 0.0)
        self.st = None
        self.pressure = np.zeros(0, np.float32)
        self.uid_sorted = None

    def enable_iisph(self, group, normal):
        self.group_g = np.asarray(group, np.int32)
        self.normal_g = np.asarray(normal, np.float32).reshape(-1, 3)
        self.pressure = np.zeros(self.n, np.float32)

    def upload(self, sub):
        super().upload(sub)
        self.pressure = np.zeros(self.n, np.float32)

    def _flush(self):
        """the particles leave the live state (after the integration, before the exchange)"""
        if self.st is None:
            return
        for k in ("pos", "vel", "type", "mass", "restDensity"):
            self.a[k] = self.st[k].copy()
        self.a["uniqueId"] = self.uid_sorted
        self.pressure = self.st["pressure"].copy()
        self.st.close()
        self.st = None

    def pack_carry(self, own, keep, left, right, cap):
        self._flush()
        k, l, r, lost = self._classify(own, keep, left, right)
        out = {}
        for side, mask, rng in ((-1, l, left), (1, r, right)):
            if rng[1] > rng[0]:
                m = int(mask.sum())
                assert m <= cap, "send buffer too small"
                rec = self._records(mask).view(np.uint8).reshape(-1).copy()
                out[side] = (torch.from_numpy(rec), torch.from_numpy(self.pressure[mask].copy().view(np.uint8)), m)
        self.a = {name: v[k] for name, v in self.a.items()}
        self.key = self.key[k]
        self.pressure = self.pressure[k]
        self.sorted_hash = None
        return int(k.sum()), int(lost.sum()), out

    def unpack_carry(self, records, extras, count):
        if count:
            self.unpack(records, count)
            self.pressure = np.concatenate([self.pressure, extras.numpy().view(np.float32)[:count]])

    def sort(self):
        self._flush()
        order = np.argsort(self.key.astype(np.uint64), kind="stable")
        self.a = {name: v[order] for name, v in self.a.items()}
        uid = self.a["uniqueId"]
        st = self.orc.new_state(self.n, self.num_cells)
        load = {k: self.a[k] for k in ("pos", "vel", "type", "mass", "restDensity")}
        load.update(group=self.group_g[uid], normal=self.normal_g[uid], pressure=self.pressure[order])
        st.load(load)
        st.sort_particles(self.p)
        h = st["particleHash"].copy()
        self.uid_sorted = uid[st["uniqueId"]]
        n = len(h)
        first = np.ones(n, bool)
        first[1:] = h[1:] != h[:-1]
        start = np.maximum.accumulate(np.where(first, np.arange(n), 0))
        self.key = (h.astype(np.int64) << 32) | (np.arange(n) - start)
        self.sorted_hash = h
        self.st = st

    def setup_iisph(self):
        self.sort()
        self.st.iisph_factor(self.p)
        self.st.rigid_particle_volume(self.p)

    def iisph_factor(self): self.st.iisph_factor(self.p)
    def iisph_npf(self): self.st.non_pressure_force_multiphase(self.p)
    def iisph_predict_density(self): self.st.iisph_predict_density(self.p)
    def iisph_calc_dijpjl(self): self.st.iisph_calc_dijpjl(self.p)
    def iisph_calc_new_pressure(self): self.st.iisph_calc_new_pressure(self.p)
    def iisph_update(self): self.st.iisph_update(self.p)

    def layer_list(self, lo, hi):
        layer = self._layer_of_hash(self.sorted_hash)
        return torch.from_numpy(np.nonzero((layer >= lo) & (layer < hi))[0].astype(np.int32))

    def gather(self, fields, idx):
        i = idx.numpy()
        parts = [np.ascontiguousarray(self.st[f][i]).view(np.uint8).reshape(-1) for f in fields]
        return torch.from_numpy(np.concatenate(parts) if parts else np.zeros(0, np.uint8))

    def scatter(self, fields, idx, data):
        i, raw, at = idx.numpy(), data.numpy(), 0
        for f in fields:
            w = IISPH_FIELD_WIDTH[f]
            nb = len(i) * w * 4
            vals = raw[at:at + nb].view(np.float32)
            self.st[f][i] = vals.reshape(-1, 3) if w == 3 else vals
            at += nb
        assert at == len(raw), "field refresh: the two slabs disagree on how many particles the layers hold"

    def field_bytes(self, fields, count):
        return count * sum(IISPH_FIELD_WIDTH[f] for f in fields) * 4

    def owned_error_sum(self, own):
        layer = self._layer_of_hash(self.sorted_hash)
        m = (layer < 0) | ((layer >= own[0]) & (layer < own[1]))
        return float(self.st["error"][m].astype(np.float64).sum())

    def get(self, name):
        if name == "particleHash":
            return self.sorted_hash
        if name == "uniqueId":
            return self.uid_sorted if self.st is not None else self.a["uniqueId"]
        if self.st is not None:
            return self.st[name].copy()
        return self.pressure if name == "pressure" else self.a[name]

    def close(self):
        if self.st is not None:
            self.st.close()
            self.st = None


class OracleDfsphSlabBackend(OracleIisphSlabBackend):
    """... and for the DFSPH step on slabs (fluidsolver_amd.slabs.DfsphSlabs): what rides with the records is v_star
    (three reals), kept here in the array the IISPH stand-in keeps the pressure in"""
    carry_width = 3

    def enable_dfsph(self):
        self.group_g = None
        self.pressure = np.zeros((self.n, 3), np.float32)

    def upload(self, sub):
        OracleSlabBackend.upload(self, sub)
        self.pressure = np.zeros((self.n, 3), np.float32)

    def _flush(self):
        if self.st is None:
            return
        for k in ("pos", "vel", "type", "mass", "restDensity"):
            self.a[k] = self.st[k].copy()
        self.a["uniqueId"] = self.uid_sorted
        self.pressure = self.st["v_star"].copy()
        self.st.close()
        self.st = None

    def pack_carry(self, own, keep, left, right, cap):
        kept, lost, out = super().pack_carry(own, keep, left, right, cap)
        return kept, lost, out

    def unpack_carry(self, records, extras, count):
        if count:
            self.unpack(records, count)
            self.pressure = np.concatenate([self.pressure, extras.numpy().view(np.float32)[:3 * count].reshape(-1, 3)])

    def sort(self):
        self._flush()
        order = np.argsort(self.key.astype(np.uint64), kind="stable")
        self.a = {name: v[order] for name, v in self.a.items()}
        uid = self.a["uniqueId"]
        st = self.orc.new_state(self.n, self.num_cells)
        load = {k: self.a[k] for k in ("pos", "vel", "type", "mass", "restDensity")}
        load.update(v_star=self.pressure[order])
        st.load(load)
        st.sort_particles(self.p)
        h = st["particleHash"].copy()
        self.uid_sorted = uid[st["uniqueId"]]
        n = len(h)
        first = np.ones(n, bool)
        first[1:] = h[1:] != h[:-1]
        start = np.maximum.accumulate(np.where(first, np.arange(n), 0))
        self.key = (h.astype(np.int64) << 32) | (np.arange(n) - start)
        self.sorted_hash = h
        self.st = st

    def dfsph_density_alpha(self): self.st.compute_density_alpha(self.p)
    def dfsph_npf(self): self.st.compute_non_pressure_force(self.p)
    def dfsph_solve_density_stiff(self): self.st.solve_density_stiff(self.p)
    def dfsph_solve_divergence_stiff(self): self.st.solve_divergence_stiff(self.p)
    def dfsph_apply_pstiff(self): self.st.apply_pstiff(self.p)
    def dfsph_update_position(self): self.st.update_position(self.p)
    def dfsph_update_velocities(self): self.st.update_velocities(self.p)

    def owned_error_max(self, own, start):
        layer = self._layer_of_hash(self.sorted_hash)
        m = (layer < 0) | ((layer >= own[0]) & (layer < own[1]))
        e = self.st["error"][m]
        e = e[~np.isnan(e)]
        return float(max(np.float32(start), e.max())) if len(e) else float(np.float32(start))

    def get(self, name):
        if name == "v_star" and self.st is None:
            return self.pressure
        return super().get(name)
