"""Inference entry -- the MI355X counterpart of the reference's inference.py (functions :19-165, CLI :167-192).

Same functions (`normalize, get_audio_feature, blur, process, inference_dataset, torch_transform`), same CLI
flags, same output files `save_path/<video>/<frame name>`; what differs:
  * the model forward and the map post-processing (blur -> exp -> resize -> min-max -> uint8) run on the GPU
    through the C ABI; one uint8 map comes back per frame instead of an fp32 map + five OpenCV passes;
  * sliding windows are independent, so `--batch` windows go through one forward; with `--graph` that forward + the
    post-process kernels are ONE captured hipGraph per frame shape, replayed with two batches in flight
    (`runtime.GraphPipeline`) while the host encodes the previous batch's JPEGs.  Off by default: in this loop the
    runtime's graph launch stalls for ~60 ms every third batch, which makes eager launches faster;
  * the wav is read and resampled once per video, not once per frame (inference.py:28-31 does it per window), lives on
    the GPU, and the log-spectrogram windows of a batch are ONE kernel launch (`preproc.log_spectrogram`);
  * frames are decoded on the host (PIL) and resized + normalised on the GPU with PIL's own fixed-point bilinear
    resampling (`preproc.resize_normalize`, bit-exact to PIL.Image.resize);
  * videos are sharded over ranks when launched with torch.distributed.run (one process per GPU, no collective);
  * cv2 / torchaudio / torchvision are not required: PIL does the frame decode + resize (what torchvision's
    transforms do on PIL images), scipy reads the wav, and torchaudio's sinc resampler and Spectrogram are
    restated on torch -- PARITY UNPINNED for those host-side third-party pieces;
  * `--model` selects the motion encoder (upstream: edit config.py:59) and the SyncBlock token tables are sized
    from `--resolution`, which removes the resolution/config mismatches F2 / F3 of SURVEY.md.

  python -m mspi_amd.inference --weight w.pt --path_data ./AuViDataset --dataset AVAD --model x3dl
"""
import argparse
import glob
import math
import os

import numpy as np
import torch

IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)

if __name__ == "__main__" and int(os.environ.get("WORLD_SIZE", "1")) == 1:
    # one hardware queue per in-flight graph branch (runtime.configure_hw_queues); the HIP runtime reads this when it
    # initialises, i.e. at the first HIP call below
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # 2 graphs x 3 branches + this loop's stream + the D2H copy stream

device = torch.device("cuda" if torch.cuda.device_count() > 0 else "cpu")
_RESOLUTION = [224, 384]     # (H, W) the frames are resized to; set from the CLI
_AUDIO_CACHE = {}


def normalize(img):
    img = (img - img.min()) / (img.max() - img.min())
    return img


# ----------------------------------------------------------------------------- audio front end (host)
def _sinc_resample(wave, orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99):
    """torchaudio.transforms.Resample defaults ('sinc_interp_hann'), restated: a strided conv with a
    Hann-windowed sinc kernel bank."""
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return wave
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernel = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base / orig)
    kernel = kernel.to(torch.float32)
    n = wave.shape[-1]
    x = torch.nn.functional.pad(wave[:, None], (width, width + orig))
    y = torch.nn.functional.conv1d(x, kernel, stride=orig).transpose(1, 2).reshape(wave.shape[0], -1)
    return y[..., : math.ceil(new * n / orig)]


def _load_wav_16k(audio_path):
    if audio_path not in _AUDIO_CACHE:
        from scipy.io import wavfile
        sr, data = wavfile.read(audio_path)
        if data.dtype.kind == "i":
            data = data.astype(np.float32) / float(2 ** (8 * data.dtype.itemsize - 1))
        elif data.dtype.kind == "u":
            data = (data.astype(np.float32) - 128.0) / 128.0
        wave = torch.as_tensor(np.atleast_2d(data.T if data.ndim == 2 else data), dtype=torch.float32)
        wave = _sinc_resample(wave, sr, 16000)
        if wave.shape[0] == 2:
            wave = torch.mean(wave, dim=0).unsqueeze(0)
        _AUDIO_CACHE.clear()
        _AUDIO_CACHE[audio_path] = wave
    return _AUDIO_CACHE[audio_path]


def audio_segment(n_samples, start_idx, fps, len_snippet=32, num_frames=None):
    """(start, length) in 16 kHz samples of the window that get_audio_feature cuts for the clip starting at frame
    `start_idx` (inference.py:33-41; python slicing clips the end to the wave).  Windows of 256 samples or fewer cannot be
    reflect-padded (upstream's Spectrogram raises on them): they are reported as empty = "no audio" (0.02 fill)."""
    mm = 16000
    if num_frames is not None:
        start = int(np.round((start_idx / num_frames * n_samples)))
        end = int(np.round(((start_idx + len_snippet + 1) / num_frames * n_samples)))
    else:
        start = int(np.round((start_idx / float(fps)) * mm))
        end = int(np.round(((start_idx + len_snippet + 1) / float(fps)) * mm))
    start, end = min(max(start, 0), n_samples), min(max(end, 0), n_samples)
    return (start, end - start) if end - start > 256 else (0, 0)


def get_audio_feature(audio_path, start_idx, fps, len_snippet=32, mode=False, num_frames=None):
    """Log-spectrogram window [1,257,111] for the clip starting at frame `start_idx` (inference.py:24-63).  Host form with
    the reference's signature; the clip loop uses the batched GPU form (`preproc.log_spectrogram`) instead."""
    spectro_shape = (257, 111)
    if os.path.exists(audio_path):
        audio = _load_wav_16k(audio_path)
        mm = 16000
        if num_frames is not None:
            mm = audio.shape[-1]
            start = int(np.round((start_idx / num_frames * mm)))
            end = int(np.round(((start_idx + len_snippet + 1) / num_frames * mm)))
        else:
            start = int(np.round((start_idx / float(fps)) * mm))
            end = int(np.round(((start_idx + len_snippet + 1) / float(fps)) * mm))
        audio = audio[:, start:end]
        if mode:
            audio = torch.flip(audio, [1])
        # torchaudio.transforms.Spectrogram(n_fft=512, hop_length=160): hann window, centre + reflect pad, power 2
        spec = torch.stft(audio, n_fft=512, hop_length=160, win_length=512, window=torch.hann_window(512), center=True,
                          pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
        audio = torch.log(spec.abs().pow(2.0) + 1e-6)
        means = audio.mean(dim=1, keepdim=True)
        stds = audio.std(dim=1, keepdim=True)
        aud = (audio - means) / (stds + 1e-6)
        tmp = torch.zeros(1, spectro_shape[0], spectro_shape[1]) + 0.02
        if audio.shape[-1] <= spectro_shape[1]:
            tmp[:, :, : audio.shape[-1]] = aud
            aud = tmp
        else:
            aud = aud[:, :, : spectro_shape[1]]
    else:
        aud = torch.zeros((1, spectro_shape[0], spectro_shape[1])) + 0.02
    return aud


def blur(img):
    """Host Gaussian blur 11x11 (sigma 2.0, reflect-101) of a numpy map -- what cv2.GaussianBlur(img,(11,11),0)
    computes.  `process` does not call it: the device post-process kernel blurs in place of it."""
    k = np.exp(-(np.arange(11, dtype=np.float32) - 5) ** 2 / (2 * 2.0 ** 2))
    k /= k.sum()
    p = np.pad(np.asarray(img, dtype=np.float32), 5, mode="reflect")
    p = np.stack([p[:, i:i + img.shape[1]] for i in range(11)], -1) @ k
    return np.stack([p[i:i + img.shape[0]] for i in range(11)], -1) @ k


@torch.no_grad()
def process(model, frames, frame_idx, vname, img_size, audio_feature=None, args=None, labels=None, frame_feats=None):
    """frames [B,3,T,H,W] (B sliding windows), frame_idx: list of B output file names.  Writes B maps.
    frame_feats: cached per-frame image-branch features of the windows' frames (model.encode_frames)."""
    from . import engine as E
    frames = frames.to(device, non_blocking=True)
    kw = {} if frame_feats is None else {"frame_feats": frame_feats}
    if args.use_sound:
        pred = model(frames, audio_feature.to(device, non_blocking=True), **kw)[0]
    else:
        pred = model(frames, **kw)[0]
    maps = E.postprocess_u8(pred, (img_size[1], img_size[0])).cpu().numpy()      # img_size is (W, H) as in cv2.resize
    E.check_range(sync=False)                # range guard of the f16x3 GEMMs (the .cpu() above synchronised): raise, never write NaN-as-0 maps
    os.makedirs(os.path.join(args.save_path, vname), exist_ok=True)
    names = frame_idx if isinstance(frame_idx, (list, tuple)) else [frame_idx]
    _write_maps(maps, names, vname, args)


def _write_maps(maps, names, vname, args):
    from PIL import Image
    for name, m in zip(names, maps):
        # quality=95 is cv2.imwrite's default IMWRITE_JPEG_QUALITY (the reference writes with cv2, inference.py:90); PIL's
        # own default of 75 would change every saved map and the metrics computed from the files
        Image.fromarray(m).save(os.path.join(args.save_path, vname, name), quality=95)


class _WindowRunner:
    """The clip loop's launch path: model forward + post-processing of a batch of `bs` windows as one hipGraph per input
    shape (frames are all resized to one resolution, so a run captures once), two batches in flight.  `run()` queues a
    batch and hands back the PREVIOUS batch's uint8 maps, so the host's JPEG encoding overlaps the GPU's next batch."""

    def __init__(self, model, bs, img_size, use_sound, depth=2):
        self.model, self.bs, self.img_size, self.use_sound, self.depth = model, bs, img_size, use_sound, depth
        self.pipe, self.key, self.prev = None, None, None
        self.loop_stream = None       # a stream on a hardware queue the graphs do not use, for the loop's own launches

    def _fn(self, cached):
        from . import engine as E
        model, out_hw = self.model, (self.img_size[1], self.img_size[0])      # img_size is (W, H) as in cv2.resize

        def fn(*t):
            clips, rest = t[0], list(t[1:])
            args = [clips] + ([rest.pop(0)] if self.use_sound else [])
            kw = {"frame_feats": (rest[0], rest[1])} if cached else {}
            return E.postprocess_u8(model(*args, **kw)[0], out_hw)
        return fn

    def run(self, inputs, names, vname, args, cached):
        """inputs: (clips[, audio][, f1, f0]) for n <= bs windows -> writes nothing itself; returns (maps, names, vname)
        of the previously queued batch, or None."""
        from .runtime import GraphPipeline
        n = inputs[0].shape[0]
        if n > self.bs:                     # the loop adds two windows per frame while the reversed ones last: split
            per = [t.shape[0] // n for t in inputs]
            first = self.run([t[: self.bs * k] for t, k in zip(inputs, per)], names[: self.bs], vname, args, cached)
            if first is not None:
                _write_maps(*first, args)
            return self.run([t[self.bs * k:] for t, k in zip(inputs, per)], names[self.bs:], vname, args, cached)
        if n < self.bs:                     # last, partial batch of a video: pad by repeating the final window
            T = inputs[0].shape[2]
            padded = []
            for t in inputs:
                per = t.shape[0] // n       # 1 for clips / audio, T for the (b t)-ordered frame features
                padded.append(torch.cat([t, t[-per:].repeat((self.bs - n,) + (1,) * (t.dim() - 1))]))
            inputs = padded
        key = tuple(tuple(t.shape) for t in inputs) + (cached,)
        if self.key != key:
            out = self.finish()
            if out is not None:
                _write_maps(*out, args)
            self.pipe = GraphPipeline(self._fn(cached), [t.to(device) for t in inputs], depth=self.depth, layouts=3)
            self.key = key
            idle = self.pipe.prepare(2)
            idle[1].wait_stream(torch.cuda.current_stream())
            self.loop_stream = idle[1]
        ticket = self.pipe.submit(*[t.to(device, non_blocking=True) for t in inputs])
        done, self.prev = self.prev, (ticket, n, names, vname)
        return self._collect(done)

    def _collect(self, rec):
        if rec is None:
            return None
        ticket, n, names, vname = rec
        return self.pipe.fetch_host(ticket)[:n].numpy(), names, vname    # pinned host buffer of that slot: written out before
                                                                         # the slot is fetched again

    def finish(self):
        out, self.prev = self._collect(self.prev), None
        return out


def torch_transform(path):
    """Resize to the model resolution, scale to [0,1], ImageNet-normalise (inference.py:154-165)."""
    from PIL import Image
    from . import preproc
    img = Image.open(path).convert("RGB")
    sz = img.size
    rgb = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).to(device, non_blocking=True)     # decode on the host,
    t = preproc.resize_normalize(rgb, (_RESOLUTION[0], _RESOLUTION[1]), IMAGENET_DEFAULT_MEAN, IMAGENET_DEFAULT_STD)
    return t, sz                                                       # resize + ToTensor + Normalize on the GPU


def _flush(model, batch, vname, img_size, args, feats=None, runner=None):
    if not batch:
        return
    clips = torch.stack([b[0] for b in batch])
    if torch.is_tensor(batch[0][1]):
        auds = torch.stack([b[1] for b in batch])
    else:                                   # (start, length, reversed) windows of the video's wave: one launch for the batch
        from . import preproc
        wave = batch[0][4]
        if wave is None:                    # no wav file: the constant the reference feeds (inference.py:60-61)
            auds = torch.full((len(batch), 1, 257, 111), 0.02, device=device)
        else:
            auds = preproc.log_spectrogram(wave, [b[1] for b in batch], 111)
    ff = None
    if feats is not None:     # (b t) order: the windows' frame indices, reversed for the time-reversed windows
        idx = [j for b in batch for j in b[3]]
        ff = (torch.stack([feats[j][0] for j in idx]), torch.stack([feats[j][1] for j in idx]))
    if runner is None:
        process(model, clips, [b[2] for b in batch], vname, img_size, audio_feature=auds, args=args, frame_feats=ff)
    else:
        inputs = [clips] + ([auds] if args.use_sound else []) + (list(ff) if ff is not None else [])
        out = runner.run(inputs, [b[2] for b in batch], vname, args, ff is not None)
        if out is not None:
            _write_maps(*out, args)
    batch.clear()


class _FrameFeatureCache:
    """Per-frame image-branch features of one video.  A stride-1 window shares 15 of its 16 frames
    with its neighbour and the ConvNeXt-T + smoothing convs never mix frames, so every frame is encoded ONCE (in chunks
    of `chunk` frames, ahead of the window that first needs it) instead of 16 times: 58 % of the model's FLOPs drop to
    1/16.  Entries older than the oldest frame still in a window are dropped."""

    def __init__(self, model, load_frame, n_frames, chunk=16):
        self.model, self.load, self.n, self.chunk = model, load_frame, n_frames, chunk
        self.feats, self.next = {}, 0

    def upto(self, i):
        while self.next <= i:
            hi = min(self.n, self.next + self.chunk)
            frames = torch.stack([self.load(j) for j in range(self.next, hi)]).to(device, non_blocking=True)
            f1, f0 = self.model.encode_frames(frames)
            for j in range(self.next, hi):
                self.feats[j] = (f1[j - self.next], f0[j - self.next])
            self.next = hi

    def drop_before(self, i):
        for j in [j for j in self.feats if j < i]:
            del self.feats[j]


def inference_dataset(model, args):
    """Sliding 16-frame window, stride 1; the first 15 frames come from the time-reversed first windows
    (inference.py:94-152).  The loop's own launches (frame resize, spectrogram windows, per-frame feature cache) go to a
    non-blocking side stream: on torch's default stream -- HIP's NULL stream -- each of them would be an implicit barrier
    against the hipGraph batches in flight (runtime.py)."""
    if device.type != "cuda":
        raise RuntimeError("mspi_amd.inference needs an MI355X (no CPU fallback)")
    side = torch.cuda.Stream(device)
    side.wait_stream(torch.cuda.current_stream(device))
    with torch.cuda.stream(side):
        _inference_dataset(model, args)
    torch.cuda.current_stream(device).wait_stream(side)


def _inference_dataset(model, args):
    len_temporal = args.clip_size
    if args.dataset == "DIEM":
        file_name = "DIEM_list_test_fps.txt"
    else:
        file_name = "{}_list_test_{}_fps.txt".format(args.dataset, args.split)
    list_data, videos_fps = [], {}
    with open(os.path.join(args.path_data, "fold_lists", file_name), "r") as f:
        for line in f.readlines():
            name, frame_num, fps = line.split(" ")
            list_data.append(name)
            videos_fps[name] = fps
    list_data.sort()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    list_data = list_data[rank::world]            # videos are independent units: shard, no collective
    print(list_data)
    bs = max(1, getattr(args, "batch", 1))
    runner = _WindowRunner(model, bs, (640, 480), args.use_sound) if getattr(args, "graph", False) else None
    for vname in list_data:
        print("Processing: " + vname)
        audio_path = os.path.join(args.path_data, "video_audio", args.dataset, vname, vname + ".wav")
        list_frames = glob.glob(os.path.join(args.path_data, "video_frames", args.dataset, vname, "*.jpg"))
        list_frames.sort(key=lambda x: int(os.path.basename(x).split(".")[0].split("_")[1]))
        os.makedirs(os.path.join(args.save_path, vname), exist_ok=True)
        if len(list_frames) < 2 * len_temporal - 1:
            print("More frames are needed")
            continue
        snippet, batch = [], []
        wave = _load_wav_16k(audio_path).reshape(-1).to(device) if os.path.exists(audio_path) else None   # once per video
        n_wave = 0 if wave is None else wave.numel()
        img_size = (640, 480)
        loaded = {}

        def load_frame(j):
            if j not in loaded:
                loaded[j] = torch_transform(list_frames[j])[0]
            return loaded[j]

        cache = None
        if getattr(args, "cache_frames", True) and hasattr(model, "encode_frames"):
            cache = _FrameFeatureCache(model, load_frame, len(list_frames))
        for i in range(len(list_frames)):
            if runner is not None and runner.loop_stream is not None and torch.cuda.current_stream() != runner.loop_stream:
                runner.loop_stream.wait_stream(torch.cuda.current_stream())
                torch.cuda.set_stream(runner.loop_stream)         # from here on the loop launches on a free hardware queue
            snippet.append(load_frame(i))
            if i >= len_temporal - 1:
                first = i - len_temporal + 1
                if cache is not None:
                    cache.upto(i)
                clip = torch.stack(snippet).permute(1, 0, 2, 3)          # [3,T,H,W]
                st, ln = audio_segment(n_wave, first, videos_fps[vname]) if wave is not None else (0, 0)
                batch.append((clip, (st, ln, 0), os.path.basename(list_frames[i]), list(range(first, i + 1)), wave))
                if i < 2 * len_temporal - 2:      # first (len_temporal-1) frames: reversed clip + reversed audio
                    batch.append((torch.flip(clip, [1]), (st, ln, 1), os.path.basename(list_frames[first]),
                                  list(range(i, first - 1, -1)), wave))
                if len(batch) >= bs:
                    _flush(model, batch, vname, img_size, args, cache.feats if cache else None, runner)
                    if cache is not None:
                        cache.drop_before(first + 1)
                del snippet[0]
                for j in [j for j in loaded if j <= first and (cache is None or j < cache.next)]:
                    del loaded[j]
        _flush(model, batch, vname, img_size, args, cache.feats if cache else None, runner)
        if runner is not None:               # the last batch in flight
            out = runner.finish()
            if out is not None:
                _write_maps(*out, args)
    torch.cuda.current_stream().synchronize()   # whichever stream the loop ended on (it moves to an idle hardware queue)


def build_model(model_name, resolution, wa=111, weight=None, use_sound=True):
    """cfg for `model_name` with the SyncBlock tables sized for `resolution` / a 257 x wa spectrogram."""
    from . import testing as T
    from .model.model_utils import AudioVisualSaliencyModel, VisualSaliencyModel
    t_tok = {"x3dl": 16, "slowfast4x16": 4, "s3d": 4}.get(model_name, 8)
    cfg = T.make_cfg(model_name, num_aud_tokens=9 * ((wa + 31) // 32),
                     num_vis_tokens=t_tok * (resolution[0] // 32) * (resolution[1] // 32))
    cfg.DATA.RESOLUTION = tuple(resolution)
    model = (AudioVisualSaliencyModel if use_sound else VisualSaliencyModel)(cfg=cfg)
    if weight is not None and os.path.exists(weight):
        model.load_state_dict(torch.load(weight, map_location="cpu"), strict=False)
    from . import engine as E
    E.autotune(True)       # first forward times the conv kernel instantiations per shape (cudnn.benchmark upstream, :189)
    return model.to(device).eval()


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--weight", default="./output/mvitv2_small_224_384_16_s2.pt", type=str)
    parser.add_argument("--save_path", default="./output", type=str)
    parser.add_argument("--split", default=2, type=int)
    parser.add_argument("--path_data", default="./AuViDataset", type=str)
    parser.add_argument("--dataset", default="AVAD", type=str)
    parser.add_argument("--clip_size", default=16, type=int)
    parser.add_argument("--use_sound", default=True, type=bool)
    parser.add_argument("--model", default=os.environ.get("MSPI_MOTION_ENCODER", "mvitv2s"), type=str)
    parser.add_argument("--resolution", default=[224, 384], type=int, nargs=2, help="H W the frames are resized to")
    parser.add_argument("--batch", default=8, type=int, help="sliding windows per forward")
    parser.add_argument("--graph", dest="graph", action="store_true",
                        help="replay one hipGraph per batch of windows, two batches in flight, instead of launching eagerly")
    parser.add_argument("--no_frame_cache", dest="cache_frames", action="store_false",
                        help="re-encode all 16 frames of every window with the image encoder, as upstream does")
    args = parser.parse_args()
    print(args)
    os.makedirs(args.save_path, exist_ok=True)
    if not torch.cuda.is_available():
        raise SystemExit("mspi_amd.inference needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    _RESOLUTION[:] = args.resolution
    model = build_model(args.model, args.resolution, weight=args.weight, use_sound=args.use_sound)
    inference_dataset(model, args)
