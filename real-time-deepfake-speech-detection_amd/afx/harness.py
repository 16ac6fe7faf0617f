"""The callers either side of the model forward, mirrored so that the reference's
scoring / eval loops are a drop-in:

  produce_evaluation_file  main.py:199-221 (same signature, same "<utt> <score>" lines)
  evaluate                 trainer.py:85-132 (loss / accuracy loop body)
  calculate_EER            trainer.py:134-139
  PreEmphasis              data/preprocess.py:8-29 (stand-alone HIP kernel)
  f_state_dict_wrapper     utils.py:13-43
  adjust_duration          data/test_set.py:201-227 (tile + crop policy)
  produce_evaluation_file_distributed  -- the multi-GPU form: utterances sharded over
      ranks, one RCCL all-gather of (index, score) pairs, rank 0 writes the file in
      dataset order.
"""
import os
from collections import OrderedDict

import torch
from torch import nn

from . import dist as adist
from ._lib import call_on, check, lib, ptr, stream_ptr


def f_state_dict_wrapper(state_dict, data_parallel=False):
    """utils.py:13-43: add (data_parallel=True) or strip the ``module.`` key prefix."""
    out = OrderedDict()
    for k, v in state_dict.items():
        if data_parallel:
            out[k if k.startswith("module") else "module." + k] = v
        else:
            out[k[7:] if k.startswith("module") else k] = v
    return out


class PreEmphasis(nn.Module):
    """data/preprocess.py:8-29.  ``forward`` keeps (B,L) also at B=1 and does not print
    (SURVEY.md Q6); bypassed when ``is_pre_emphasis`` is false (:19-20)."""

    def __init__(self, device=None, sys_config=None, exp_config=None, coef=None, enabled=None):
        super().__init__()
        self.coef = float(coef if coef is not None else getattr(exp_config, "pre_emphasis", 0.97))
        self.enabled = bool(enabled if enabled is not None else getattr(exp_config, "is_pre_emphasis", True))

    def forward(self, x):
        if not self.enabled:
            return x
        if not x.is_cuda:
            raise RuntimeError("PreEmphasis: input must be on the GPU (no CPU fallback)")
        x = x.to(torch.float32).contiguous()
        y = torch.empty_like(x)
        check(call_on(x, lib().afx_k_pre_emphasis, ptr(x), x.shape[0], x.shape[1], self.coef, ptr(y)))
        return y


def adjust_duration(x, duration):
    """data/test_set.py:201-227: repeat short clips (whole copies + residue), keep the
    first ``duration`` samples.  Pure indexing on the loader side."""
    x = x.reshape(-1)
    n = x.shape[0]
    if n < duration:
        parts = [x] * (duration // n)
        if duration % n > 0:
            parts.append(x[: duration % n])
        x = torch.cat(parts, dim=0)
    return x[:duration]


def batch_adjust_duration(clips, duration, starts=None, device="cuda"):
    """The same policy for a whole ragged batch as ONE device op: ``clips`` is a list of 1-D waveforms
    of any lengths, the result a (B, duration) fp32 batch on the GPU.  ``starts`` (optional, one per
    clip) are the crop starts of data/test_set.py:229-248 for clips longer than ``duration``."""
    import ctypes as C

    from ._lib import call_on, check, lib, ptr, stream_ptr
    lens = [int(c.numel()) for c in clips]
    if not clips or min(lens) <= 0:
        raise ValueError("every clip needs at least one sample")
    offs = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int64)
    packed = torch.cat([c.reshape(-1).to(torch.float32) for c in clips]).to(device)
    offs_d = offs.to(device)
    st = None
    if starts is not None:
        for s0, n in zip(starts, lens):
            if s0 < 0 or (n >= duration and s0 > n - duration) or (n < duration and s0 != 0):
                raise ValueError("crop start outside the clip")
        st = torch.tensor(list(starts), dtype=torch.int64, device=device)
    out = torch.empty(len(clips), duration, dtype=torch.float32, device=device)
    check(call_on(packed, lib().afx_k_tile_crop, ptr(packed), ptr(offs_d), ptr(st), len(clips), duration, ptr(out)))
    return out


def calculate_EER(scores, labels):
    """trainer.py:134-139."""
    from scipy.interpolate import interp1d
    from scipy.optimize import brentq
    from sklearn import metrics
    fpr, tpr, _ = metrics.roc_curve(labels, scores, pos_label=1)
    return brentq(lambda x: 1.0 - x - interp1d(fpr, tpr)(x), 0.0, 1.0) * 100


def _loader(dataset, batch_size, num_workers):
    from torch.utils import data
    return data.DataLoader(dataset, batch_size=batch_size, shuffle=False, drop_last=False, num_workers=num_workers)


def write_score_file(save_path, utt_ids, scores):
    d = os.path.dirname(save_path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(save_path, "w") as fh:
        for f, cm in zip(utt_ids, scores):
            fh.write("{} {}\n".format(f, cm))


def _as_device(device):
    """``device`` as the reference passes it -- an int rank (main.py:48), 'cuda' / 'cuda:1' / 'cpu' or a torch.device --
    as a torch.device; a CUDA device without an index means the current one."""
    d = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    if d.type == "cuda" and d.index is None and torch.cuda.is_available():
        d = torch.device("cuda", torch.cuda.current_device())
    return d


def prefetch_to_device(batches, device):
    """Iterate ``(meta, x_on_device)`` over ``(meta, x_host)`` pairs with the NEXT batch's H2D copy
    (pinned staging buffer, side stream) running under the current batch's forward.  The compute
    stream only waits on the copy's event, so a pass is bounded by max(compute, PCIe), not the sum."""
    dev = _as_device(device)
    if not torch.cuda.is_available() or dev.type != "cuda":
        for meta, x in batches:
            yield meta, x.to(dev)
        return
    # every stream below is a stream OF `dev`, whatever torch's current device is (the reference passes device=rank and
    # never calls torch.cuda.set_device, main.py:48): the copy runs on a side stream of the engine's GPU and the wait
    # goes to that GPU's compute stream
    from .engine import side_stream
    side = side_stream(dev, "copy")  # one per GPU and process (hardware queues: see engine.side_stream)
    it = iter(batches)

    def stage(item):
        meta, x = item
        with torch.cuda.stream(side):
            xd = x.pin_memory().to(dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(side)
        return meta, xd, ev

    cur = next(it, None)
    cur = stage(cur) if cur is not None else None
    while cur is not None:
        nxt = next(it, None)
        nxt = stage(nxt) if nxt is not None else None  # in flight while `cur` is consumed
        meta, xd, ev = cur
        torch.cuda.current_stream(dev).wait_event(ev)
        xd.record_stream(torch.cuda.current_stream(dev))
        yield meta, xd
        cur = nxt


def _may_overlap(model):
    """The scoring loops may call ``model.forward_overlapped(x)`` in place of the reference's ``model(batch_x)``
    (main.py:211) only where that skips nothing: ``nn.Module.__call__`` is what runs forward (pre-)hooks -- a
    ``afx.kd.ForwardHookManager`` tap, anything the user registered -- so a model with hooks goes through ``model(x)``."""
    from torch.nn.modules import module as _m
    if getattr(model, "afx_arch", None) not in ("xlsr_aasist", "conformer") or not hasattr(model, "forward_overlapped"):
        return False
    if model._forward_hooks or model._forward_pre_hooks or _m._global_forward_hooks or _m._global_forward_pre_hooks:
        return False
    return True


def produce_evaluation_file(dataset, model, device, save_path, batch_size, num_workers=4):
    """main.py:199-221.  Scores stay on the GPU until the end of the pass (one D2H copy
    instead of the reference's per-batch ``.cpu()``), and the next batch's H2D copy overlaps the
    current forward (the reference's ``batch_x.to(device)`` is blocking)."""
    model.eval()
    names, chunks = [], []
    # two batches in flight (the back-end of a batch under the next batch's trunk, or whole forwards on alternating streams: +5 % for
    # the student, +14 % for the teacher): the scores are only read after the last batch, so nothing waits inside the loop
    overlapped, asked = _may_overlap(model), False
    with torch.no_grad():
        loader = ((utt_id, batch_x) for utt_id, batch_x, _label in _loader(dataset, batch_size, num_workers))
        for utt_id, x in prefetch_to_device(loader, device):
            if overlapped and not asked:
                # which way of keeping two batches in flight is the fastest HERE (one stream / back-end beside the next trunk / whole
                # forwards on alternating streams) is timed once on the first batch (Engine.overlap_pays); same bits either way
                overlapped, asked = bool(model.overlap_pays(x)) if hasattr(model, "overlap_pays") else overlapped, True
            out = model.forward_overlapped(x) if overlapped else model(x)
            chunks.append(out[:, 1])  # bonafide score (main.py:211-212)
            names.extend(utt_id)
        if overlapped:
            model.join_overlapped()
    _check_finite(model)  # (an fp16 operand overflow is an error here, not a NaN -- or a plausible-looking -- score in the file)
    scores = torch.cat(chunks).cpu().numpy().ravel().tolist() if chunks else []
    write_score_file(save_path, names, scores)
    return names, scores


def _check_finite(model):
    if hasattr(model, "check_finite"):
        model.check_finite()


def checkpoint_comment(ckpt_path):
    """main.py:265-266: the score-file suffix of a swept checkpoint = the last three '_'-separated fields of its path."""
    parts = ckpt_path.split("_")
    return parts[-3] + "_" + parts[-2] + "_" + parts[-1]


def score_tracks(model, tracks, datasets, save_paths, device, batch_size, comment=None, num_workers=4, log=print):
    """The track loop of main.py:406-451 (dup :296-371): for every requested track, in order -- skip it when its score
    file already exists (":File existed, skip"), else build its dataset and call produce_evaluation_file.
    ``datasets``: track -> zero-argument callable returning the Dataset (the reference constructs
    ASVspoof2019LA_eval / ASVspoof2021DF_eval / ... there; datasets are the caller's, SURVEY 8 out of scope);
    ``save_paths``: track -> score file ('.txt' gets ``_<comment>`` inserted as at main.py:398-404).  An 'InTheWild'
    track without its own path uses the DF21 path with the track name substituted (main.py:440-441).  Returns
    {track: path} of the files written."""
    paths = dict(save_paths)
    if comment is not None:
        paths = {t: p.replace(".txt", f"_{comment}.txt") for t, p in paths.items()}
    done = {}
    for track in tracks:
        if track not in paths and "DF21" in paths:
            paths[track] = paths["DF21"].replace("DF21", track)
        if track not in datasets or track not in paths:
            raise ValueError(f"Track {track} not found.")
        log(f"Evaluating {track}")
        if os.path.exists(paths[track]):
            log("File existed, skip")
            continue
        produce_evaluation_file(datasets[track](), model, device, paths[track], batch_size, num_workers=num_workers)
        done[track] = paths[track]
    return done


def score_all_checkpoints(folder, build_model, tracks, datasets, save_paths, device, batch_size, num_workers=4, log=print):
    """``--score_all_folder_path`` (main.py:258-371): every '*.pt*' file of ``folder`` is loaded into a fresh model the way
    the reference does -- through a DataParallel-style 'module.' prefix (main.py:288-291, utils.py:13-43) -- and scored
    on every track, the score files suffixed with the checkpoint's comment.  ``build_model()`` returns the model on
    ``device``.  Returns {checkpoint: {track: path}}."""
    out = {}
    for name in sorted(n for n in os.listdir(folder) if ".pt" in n):
        ckpt = os.path.join(folder, name)
        model = build_model()
        sd = torch.load(ckpt, map_location=device)
        wrapped = f_state_dict_wrapper(sd, data_parallel=True)            # what DataParallel(model).load_state_dict sees
        model.load_state_dict(f_state_dict_wrapper(wrapped, data_parallel=False))  # ... and model = model.module
        log(f"Load checkpoint from {ckpt}")
        out[ckpt] = score_tracks(model, tracks, datasets, save_paths, device, batch_size, comment=checkpoint_comment(ckpt),
                                 num_workers=num_workers, log=log)
    return out


def ragged_collate(batch):
    """utils.py:102-106 (``my_collate``) for (utt_id, waveform, label) samples: lists, no stacking -- the clips keep
    their own lengths."""
    return [b[0] for b in batch], [b[1] for b in batch], [b[2] for b in batch]


def produce_evaluation_file_ragged(dataset, model, device, save_path, batch_size, num_workers=4):
    """main.py:199-221 for UN-CROPPED clips: batches of different lengths go through ``model.forward_ragged`` (each clip
    scored exactly as if alone), one score line per utterance in dataset order."""
    from torch.utils import data
    model.eval()
    names, chunks = [], []
    loader = data.DataLoader(dataset, batch_size=batch_size, shuffle=False, drop_last=False, num_workers=num_workers,
                             collate_fn=ragged_collate)
    with torch.no_grad():
        for utt_id, clips, _label in loader:
            chunks.append(model.forward_ragged([torch.as_tensor(c) for c in clips])[:, 1])
            names.extend(utt_id)
    _check_finite(model)
    scores = torch.cat(chunks).cpu().numpy().ravel().tolist() if chunks else []
    write_score_file(save_path, names, scores)
    return names, scores


def evaluate(model, loader, device, loss_fn=None, preprocessor=None):
    """trainer.py:85-132: (mean loss, accuracy %) over a loader."""
    model.eval()
    n_total, n_correct, loss_sum = 0, 0, 0.0
    with torch.no_grad():
        for _utt, x, label in loader:
            x = x.to(device)
            label = label.view(-1).type(torch.int64).to(device)
            if preprocessor is not None:
                x = preprocessor(x)
            out = model(x)
            if loss_fn is not None:
                loss_sum += loss_fn(out, label).item() * x.size(0)
            n_correct += (out.max(dim=1)[1] == label).sum().item()
            n_total += x.size(0)
    _check_finite(model)
    return loss_sum / max(n_total, 1), n_correct / max(n_total, 1) * 100


class _Shard(torch.utils.data.Dataset):
    """This rank's utterances; every item keeps its dataset index AND its utterance id, so that nobody ever has to
    call ``dataset[i]`` again (the reference datasets decode the audio file in ``__getitem__``)."""

    def __init__(self, base, idx):
        self.base, self.idx = base, [int(i) for i in idx if i >= 0]

    def __len__(self):
        return len(self.idx)

    def __getitem__(self, i):
        utt, x, label = self.base[self.idx[i]]
        return self.idx[i], utt, x, label


def produce_evaluation_file_distributed(dataset, model, device, save_path, batch_size, num_workers=4, group=None):
    """Multi-GPU scoring: rank r scores utterances r, r+W, ...; ONE all-gather of
    (index, score) pairs (RCCL when the group's backend is nccl); rank 0 writes the file
    in dataset order.  Returns (indices, scores) on every rank.

    The utterance ids are collected in the scoring loop (each clip is decoded exactly once, by the loader's workers)
    and gathered as Python objects after the scores -- rank 0 never re-reads the dataset for a name."""
    import contextlib
    import torch.distributed as tdist
    dev = _as_device(device)
    # the engine's GPU is made current for the whole pass: under NCCL / RCCL all_gather_object stages its byte tensors on
    # torch.cuda.current_device(), which without this is cuda:0 on every rank of a caller that passes device=rank
    with (torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()):
        return _produce_evaluation_file_distributed(dataset, model, dev, save_path, batch_size, num_workers, group)


def _produce_evaluation_file_distributed(dataset, model, device, save_path, batch_size, num_workers, group):
    import torch.distributed as tdist
    rank, world = tdist.get_rank(group), tdist.get_world_size(group)
    idx = adist.shard_indices(len(dataset), rank, world)
    model.eval()
    scores = torch.zeros(idx.numel(), dtype=torch.float32, device=device)
    names, outs = {}, []
    overlapped, asked = _may_overlap(model), False
    with torch.no_grad():
        loader = (((i, utt), x) for i, utt, x, _label in _loader(_Shard(dataset, idx.tolist()), batch_size, num_workers))
        for (i, utt), x in prefetch_to_device(loader, device):
            if overlapped and not asked:
                # next to a process group's streams the two-stream form is not taken on faith: timed once per engine on
                # the first batch (Engine.overlap_pays), the faster form is what the pass issues
                overlapped, asked = bool(model.overlap_pays(x)) if hasattr(model, "overlap_pays") else overlapped, True
            outs.append(model.forward_overlapped(x) if overlapped else model(x))  # (graph back-end under the next batch's trunk)
            names.update(zip((int(v) for v in (i.tolist() if torch.is_tensor(i) else i)), utt))
        if overlapped:
            model.join_overlapped()
    _check_finite(model)
    if outs:
        got = torch.cat([o[:, 1] for o in outs])
        scores[: got.numel()] = got
    gi, gs = adist.all_gather_scores(idx.to(device=device, dtype=torch.int32), scores, world, group)
    mi, ms = adist.merge_scores(gi, gs)
    all_names = [None] * world
    tdist.all_gather_object(all_names, names, group=group)
    if rank == 0:
        lut = {}
        for d in all_names:
            lut.update(d)
        write_score_file(save_path, [lut[int(i)] for i in mi.tolist()], ms.cpu().numpy().ravel().tolist())
    return mi, ms
