"""Extraction -> frame-to-frame SearchByProjection on a batch of consecutive frames, with every
buffer resident on the GPU: orb_extract_batch_device followed by
orb_search_by_projection_frames_device on the extractor's own output records (the work that
Frame::ExtractORB, reference src/Frame.cc:220-228, and Tracking::TrackWithInitialPose,
src/Tracking.cc:1065-1067, do one frame at a time).  The benchmark times this chain and the
parity tests compare it with a CPU restatement of the reference; torch only provides the device memory."""
import numpy as np

from . import _lib
from ._lib import KP_DTYPE, FrameBounds, check
from .extractor import ORBextractor
from .synth import SequenceScene


class DeviceChain:
    def __init__(self, rows, cols, n_frames, nfeatures=2000, nlevels=8, device=0, nsets=1, th=7.0,
                 check_orientation=True, n_handles=1):
        import torch
        self.torch = torch
        self.rows, self.cols, self.B, self.local = int(rows), int(cols), int(n_frames), int(device)
        self.dev = torch.device("cuda", self.local)
        self.L = _lib.lib()
        # n_handles extractor handles: a handle serves one stream at a time, so batches that are to be
        # in flight together (step k + 1 beside step k) take one each
        self.exs = [ORBextractor(nfeatures, 1.2, nlevels, 20, 7, device=self.local) for _ in range(max(1, n_handles))]
        self.ex = self.exs[0]
        self.nlevels = nlevels
        self.max_kp = self.ex.max_keypoints(rows, cols)
        self.scene = SequenceScene(rows, cols)
        self.th, self.check_ori = float(th), int(bool(check_orientation))
        B, mk, dev = self.B, self.max_kp, self.dev
        # nsets sets of extractor outputs: a matcher may read one while the next extraction fills another
        self.kps = [torch.zeros((B, mk, 6), dtype=torch.float32, device=dev) for _ in range(nsets)]
        self.desc = [torch.zeros((B, mk, 32), dtype=torch.uint8, device=dev) for _ in range(nsets)]
        self.nout = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(nsets)]
        self.npairs = B - 1
        npairs = max(self.npairs, 1)
        self.sf = self.ex.GetScaleFactors()
        self.scratch_bytes = self.L.orb_match_scratch_bytes(npairs, mk)
        self.d_scratch = torch.zeros(self.scratch_bytes, dtype=torch.uint8, device=dev)
        self.d_match = torch.zeros((npairs, mk), dtype=torch.int32, device=dev)
        self.d_nm = torch.zeros(npairs, dtype=torch.int32, device=dev)
        self.d_cur_obs = torch.full((npairs, mk), -1, dtype=torch.int32, device=dev)
        self.bounds = FrameBounds(*self.scene.bounds)
        self.d_img = None
        self.d_xw = self.d_last_obs = self.d_tcw = None

    def load(self, seq, first_frame=0):
        """Uploads the frames (B, rows, cols) uint8, extracts them once and builds the map points
        of every frame from its keypoints and the poses of the pairs (frame p, frame p + 1);
        `first_frame` is the sequence index of seq[0]."""
        torch = self.torch
        assert seq.shape == (self.B, self.rows, self.cols) and seq.dtype == np.uint8
        self.d_img = torch.from_numpy(np.ascontiguousarray(seq)).to(self.dev)
        self.extract(0)
        torch.cuda.synchronize()
        if self.npairs < 1:
            return
        kps = self.kps[0][:self.npairs].cpu().numpy()
        nout = self.nout[0][:self.npairs].cpu().numpy()
        xw = np.zeros((self.npairs, self.max_kp, 3), np.float32)
        obs = np.full((self.npairs, self.max_kp), -1, np.int32)
        tcw = np.zeros((self.npairs, 16), np.float32)
        for p in range(self.npairs):
            n = min(int(nout[p]), self.max_kp)
            xw[p, :n] = self.scene.map_points(kps[p, :n, 0], kps[p, :n, 1], first_frame + p)
            obs[p, :n] = 1
            tcw[p] = self.scene.tcw(first_frame + p + 1).reshape(16)
        self.d_xw = torch.from_numpy(xw).to(self.dev)
        self.d_last_obs = torch.from_numpy(obs).to(self.dev)
        self.d_tcw = torch.from_numpy(tcw).to(self.dev)

    def extract(self, s=0, stream=None, handle=0):
        st = self.torch.cuda.current_stream().cuda_stream if stream is None else stream
        self.exs[handle].extract_batch_device(self.d_img.data_ptr(), self.B, self.rows, self.cols, self.kps[s].data_ptr(),
                                     self.desc[s].data_ptr(), self.nout[s].data_ptr(), self.max_kp, stream=st)

    def match(self, s=0, stream=None):
        """SearchByProjection(Cur = frame p + 1, Last = frame p, th, mono) for p = 0 .. B - 2 on
        output set s: the records of frame p + 1 are those of frame p moved on by one frame."""
        st = self.torch.cuda.current_stream().cuda_stream if stream is None else stream
        sc, mk = self.scene, self.max_kp
        n, k, d = self.nout[s].data_ptr(), self.kps[s].data_ptr(), self.desc[s].data_ptr()
        check(self.L.orb_search_by_projection_frames_device(
            self.local, self.npairs, mk, mk, self.d_tcw.data_ptr(), float(sc.fx), float(sc.fy), float(sc.cx),
            float(sc.cy), self.bounds, self.sf.ctypes.data, self.nlevels,
            n, k, self.d_last_obs.data_ptr(), self.d_xw.data_ptr(), d,
            n + 4, k + mk * KP_DTYPE.itemsize, d + mk * 32, self.d_cur_obs.data_ptr(),
            self.th, self.check_ori, self.d_scratch.data_ptr(), self.scratch_bytes, self.d_match.data_ptr(),
            self.d_nm.data_ptr(), st))

    def results(self, s=0):
        """(n_out[B], keypoints[B][max_kp] records, descriptors, cur_match[npairs][max_kp], n_matches[npairs])
        as host arrays (waits for the device)."""
        self.torch.cuda.synchronize()
        kps = self.kps[s].cpu().numpy().view(np.uint8).reshape(self.B, self.max_kp, 24).copy().view(KP_DTYPE)[..., 0]
        return (self.nout[s].cpu().numpy(), kps, self.desc[s].cpu().numpy(), self.d_match.cpu().numpy(),
                self.d_nm.cpu().numpy())
