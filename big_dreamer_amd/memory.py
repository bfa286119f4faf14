"""ExperienceReplay -- same surface and sampling semantics as the reference (src/memory.py:8-104), with
the storage mirrored in HBM and the batch gather done by a HIP kernel (bd_replay_gather), so that
``sample`` returns device tensors without a host-side gather or a bulk H2D copy.

Pixel observations are kept as uint8 (5-bit quantised frames, as the reference stores them) and de-quantised on
the device by bd_replay_gather_pixels; the dequantisation noise is drawn by torch's device generator."""
from __future__ import annotations

import numpy as np
import torch

from . import _cabi as cabi


class ExperienceReplay:
    def __init__(self, size, action_size, bit_depth, pixel_observation, observation_size, device):
        self.device = torch.device(device)
        self.size = size
        self.pixel_observation = pixel_observation
        self.bit_depth = bit_depth
        if pixel_observation:
            self.observations = np.empty((size, 3, 64, 64), dtype=np.uint8)
        else:
            self.observations = np.empty((size, observation_size), dtype=np.float32)
        self.actions = np.empty((size, action_size), dtype=np.float32)
        self.rewards = np.empty((size,), dtype=np.float32)
        self.nonterminals = np.empty((size, 1), dtype=np.float32)
        self.idx = 0
        self.full = False
        self.steps, self.episodes = 0, 0
        self._dev = None          # device mirror, created lazily / refreshed by sync_device()
        self._dirty = True
        self._out_ring, self._out_i = {}, {}
        self._pix_noise = None
        self._pix_seed, self._pix_step = None, 0
        self._ring = []           # pinned staging buffers for the index upload (async H2D, no host stall)
        self._ring_i = 0

    # -- reference semantics (src/memory.py:33-49) --
    def append(self, observation, action, reward, done):
        to_np = lambda x: x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)
        if self.pixel_observation:
            # postprocess_observation (src/utils.py:320-337): [-0.5, 0.5] float -> quantised uint8
            o = to_np(observation)
            self.observations[self.idx] = np.clip(np.floor((o + 0.5) * 2 ** self.bit_depth) * 2 ** (8 - self.bit_depth),
                                                  0, 2 ** 8 - 1).astype(np.uint8)
        else:
            self.observations[self.idx] = to_np(observation)
        self.actions[self.idx] = to_np(action)
        self.rewards[self.idx] = reward
        self.nonterminals[self.idx] = not done
        if self._dev is not None and not self._dirty:
            i = self.idx
            self._dev["observations"][i].copy_(torch.from_numpy(self.observations[i]))
            self._dev["actions"][i].copy_(torch.from_numpy(self.actions[i]))
            self._dev["rewards"][i] = float(self.rewards[i])
            self._dev["nonterminals"][i] = float(self.nonterminals[i, 0])
        self.idx = (self.idx + 1) % self.size
        self.full = self.full or self.idx == 0
        self.steps, self.episodes = self.steps + 1, self.episodes + (1 if done else 0)

    def _sample_idx(self, L):
        """src/memory.py:51-68: uniform start, rejected if the chunk crosses the write head."""
        valid_idx = False
        while not valid_idx:
            idx = np.random.randint(0, self.size if self.full else self.idx - L)
            idxs = np.arange(idx, idx + L) % self.size
            valid_idx = not self.idx in idxs[1:]
        return idxs

    def sync_device(self):
        """(Re)upload the whole buffer to HBM; afterwards append() keeps the mirror in step."""
        self._dev = {k: torch.from_numpy(getattr(self, k)).to(self.device) for k in
                     ("observations", "actions", "rewards", "nonterminals")}
        self._dirty = False

    def mark_dirty(self):
        """Call after writing the numpy arrays directly (e.g. bulk synthetic fill)."""
        self._dirty = True

    def _upload_indices(self, vec: np.ndarray) -> torch.Tensor:
        """Async H2D of the gather indices through a ring of pinned buffers, so that the host can run
        ahead of the GPU (a pageable copy would wait for all queued kernels of the previous step)."""
        if not self._ring or self._ring[0][0].numel() != vec.size:
            self._ring = [(torch.empty(vec.size, dtype=torch.int64).pin_memory(),
                           torch.empty(vec.size, dtype=torch.int64, device=self.device),
                           torch.cuda.Event()) for _ in range(4)]
            self._ring_i = 0
        pinned, dev, ev = self._ring[self._ring_i]
        self._ring_i = (self._ring_i + 1) % len(self._ring)
        ev.synchronize()                       # the copy that last used this pinned buffer has completed
        pinned.copy_(torch.from_numpy(vec))
        dev.copy_(pinned, non_blocking=True)
        ev.record()
        return dev

    def _out(self, key: str, numel: int) -> torch.Tensor:
        """Output buffers come from a ring of four persistent allocations per array, so that the kernels downstream
        see a small repeating set of operand addresses (descriptor tables and graphs can be reused) while a batch
        stays valid for the next three `sample` calls -- the engine's pipeline holds one for at most two."""
        ring = self._out_ring.setdefault((key, numel), [])
        if len(ring) < 4:
            ring.append(torch.empty(numel, dtype=torch.float32, device=self.device))
            return ring[-1]
        i = self._out_i.get((key, numel), 0)
        self._out_i[(key, numel)] = (i + 1) % 4
        return ring[i]

    def sample(self, n, L):
        """Time-major batch [obs (L,n,O), actions (L,n,A), rewards (L,n), nonterminals (L,n,1)] on the device
        (src/memory.py:70-104)."""
        idxs = np.asarray([self._sample_idx(L) for _ in range(n)])
        vec = np.ascontiguousarray(idxs.transpose().reshape(-1)).astype(np.int64)
        if self.device.type != "cuda":
            raise RuntimeError("ExperienceReplay.sample: the HIP path needs a GPU device (no CPU fallback)")
        if self._dev is None or self._dirty:
            self.sync_device()
        vidx = self._upload_indices(vec)
        out = []
        if self.pixel_observation:
            src = self._dev["observations"]
            pixels = 3 * 64 * 64
            # dequantisation noise (rand_like, src/utils.py:317) drawn inside the gather kernel: Philox keyed by the torch
            # seed at the first sample, counter = sample index (no noise tensor, no library RNG launch per step)
            if self._pix_seed is None:
                self._pix_seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
            dst = self._out("pixels", L * n * pixels)
            cabi.check(cabi.lib.bd_replay_gather_pixels_rng(src.data_ptr(), vidx.data_ptr(), L * n, pixels, self.bit_depth,
                                                            self._pix_seed, self._pix_step, dst.data_ptr(), cabi.stream()))
            self._pix_step += 1
            out.append(dst.view(L, n, 3, 64, 64))
        for key, shape in (("observations", (L, n, -1)), ("actions", (L, n, -1)), ("rewards", (L, n)),
                           ("nonterminals", (L, n, 1))):
            if key == "observations" and self.pixel_observation:
                continue
            src = self._dev[key]
            width = src.numel() // src.shape[0]
            dst = self._out(key, L * n * width)
            cabi.check(cabi.lib.bd_replay_gather(src.data_ptr(), vidx.data_ptr(), L * n, width, dst.data_ptr(),
                                                 cabi.stream()))
            out.append(dst.view(*shape))
        return out
