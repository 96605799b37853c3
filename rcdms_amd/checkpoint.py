"""Checkpoint and output-file compatibility of the stage-2 inference driver — host code only.

The reference's driver (stage2_batchtest_rcdms_model.py) reads ONE DeepSpeed file,
`./stage2/<exp>/<weights_number>/mp_rank_00_model_states.pt`, takes its `"module"` dict and routes keys by prefix
(:225-243): `unet.*` -> the UNet3D, `seen_module.*` -> fine_stack (local_module), `unseen_module.*` -> semantic_stack
(global_module); anything else is printed and dropped.  These helpers restate that routing (with the errors made explicit instead of printed), the story
sharding across ranks (`split_list`, :58-70), the PNG grid writer (`image_grid`, :79-93) and the ARLDM h5 test-split reader
(`read_story_split` / `pick_story_frames`, :41-56,440-453,257-266), so a maintainer's script keeps its file formats when it
switches to `rcdms_amd`."""
import os

import numpy as np
import torch

STAGE2_PREFIXES = (("seen_module.", "local_module"), ("unseen_module.", "global_module"), ("unet.", "unet"))


def split_stage2_state(module_sd):
    """`torch.load(ckpt)["module"]` -> {"unet": {...}, "local_module": {...}, "global_module": {...}, "other": [keys]}.
    Keys keep their order; the prefix is stripped exactly once (the reference uses str.replace, which would also rewrite a
    later occurrence of the prefix inside a key — no reference key has one, and stripping the head is what is meant)."""
    out = {"unet": {}, "local_module": {}, "global_module": {}, "other": []}
    for k, v in module_sd.items():
        for prefix, dest in STAGE2_PREFIXES:
            if k.startswith(prefix):
                out[dest][k[len(prefix):]] = v
                break
        else:
            out["other"].append(k)
    return out


def stage2_checkpoint_path(exp_name, weights_number, root="./stage2"):
    """The path the driver builds at :225."""
    return os.path.join(root, str(exp_name), str(weights_number), "mp_rank_00_model_states.pt")


def load_stage2_checkpoint(ckpt, unet, local_module, global_module, strict=True):
    """Load a DeepSpeed stage-2 checkpoint (path, the loaded file dict, or its "module" dict) into the three modules,
    as :225-243 does.  Returns the list of keys that matched no prefix."""
    if isinstance(ckpt, (str, os.PathLike)):
        if not os.path.isfile(ckpt):
            raise FileNotFoundError(f"stage-2 checkpoint {ckpt} does not exist")
        ckpt = torch.load(ckpt, map_location="cpu")
    module_sd = ckpt["module"] if "module" in ckpt and isinstance(ckpt["module"], dict) else ckpt
    parts = split_stage2_state(module_sd)
    for name in ("unet", "local_module", "global_module"):
        if not parts[name]:
            raise KeyError(f"checkpoint holds no `{[p for p, d in STAGE2_PREFIXES if d == name][0]}*` keys")
    local_module.load_state_dict(parts["local_module"], strict=strict)
    global_module.load_state_dict(parts["global_module"], strict=strict)
    unet.load_state_dict(parts["unet"], strict=strict)
    return parts["other"]


def save_stage2_checkpoint(path, unet, local_module, global_module):
    """Write the three modules in the layout above (what DeepSpeed's `save_checkpoint` leaves for the wrapper module of
    train_stage2.py: attributes `unet`, `seen_module`, `unseen_module`)."""
    sd = {}
    for prefix, mod in (("unet.", unet), ("seen_module.", local_module), ("unseen_module.", global_module)):
        for k, v in mod.state_dict().items():
            sd[prefix + k] = v.detach().cpu()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save({"module": sd}, path)


def split_list(n, m):
    """Stories 0..n-1 dealt to m ranks in contiguous runs, the first n % m runs one longer (:58-70)."""
    q, r = divmod(n, m)
    out, start = [], 0
    for i in range(m):
        end = start + q + (1 if i < r else 0)
        out.append(list(range(start, end)))
        start = end
    return out


def image_grid(imgs, rows, cols):
    """rows x cols PNG grid of float images in [0, 1], H x W x 3 each (:79-93; values are truncated to uint8 as there)."""
    from PIL import Image
    if len(imgs) != rows * cols:
        raise AssertionError(f"{len(imgs)} images do not fill a {rows} x {cols} grid")
    tiles = [Image.fromarray((np.array(im) * 255).astype(np.uint8)) for im in imgs]
    w, h = tiles[0].size
    grid = Image.new("RGB", size=(cols * w, rows * h))
    for i, t in enumerate(tiles):
        grid.paste(t, box=(i % cols * w, i // cols * h))
    return grid


def videos_to_frames(videos):
    """`pipe(...).videos` (b, 3, f, H, W) in [0, 1] -> list over stories of lists of H x W x 3 numpy frames, the layout the
    driver turns into its per-story grids and per-frame metric PNGs (:378-401)."""
    v = videos.detach().float().cpu() if isinstance(videos, torch.Tensor) else torch.as_tensor(videos).float()
    return [[v[i, :, j].permute(1, 2, 0).numpy() for j in range(v.shape[2])] for i in range(v.shape[0])]


# ---- the ARLDM h5 story file (stage2_batchtest_rcdms_model.py:41-56,440-453) ---------------------------------------------
# Layout: group "<split>" with six datasets of one entry per story — image0 .. image4: the ENCODED bytes (PNG / JPEG, a 1-D
# uint8 array) of a 128-px-wide strip that stacks five 128 x 128 candidate frames vertically (640 x 128 x 3 after decoding);
# text: one UTF-8 byte string, the five captions joined by '|'.

def _default_image_decoder():
    """cv2.imdecode(..., IMREAD_COLOR) when OpenCV is importable (what the reference calls, :43-46: H x W x 3 uint8 in BGR
    order), else the same result through PIL (decode to RGB, reverse the channel axis)."""
    try:
        import cv2
        return lambda buf: cv2.imdecode(np.asarray(buf, dtype=np.uint8), cv2.IMREAD_COLOR)
    except ImportError:
        import io
        from PIL import Image

        def decode(buf):
            rgb = np.array(Image.open(io.BytesIO(np.asarray(buf, dtype=np.uint8).tobytes())).convert("RGB"))
            return np.ascontiguousarray(rgb[:, :, ::-1])
        return decode


def read_story_split(source, split="test", decode=None):
    """The driver's whole-split read (:440-453): -> {"image0" .. "image4": [H x W x 3 uint8 BGR array per story], "text":
    [[caption x 5] per story]} — the `dataset_dict` every spawned rank receives.
    source: a path to the ARLDM .h5 file (opened with h5py — not part of this image: ImportError says so), or any mapping
    with the same layout (an open h5py.File, a dict of arrays), which is how the tests exercise it.  decode: bytes array ->
    image (default: _default_image_decoder)."""
    opened = None
    if isinstance(source, (str, os.PathLike)):
        try:
            import h5py
        except ImportError as e:
            raise ImportError("read_story_split(path) needs h5py to open the ARLDM .h5 file (not installed here); pass an open "
                              "h5py.File / any mapping with groups '<split>/image0..4' and '<split>/text' instead") from e
        if not os.path.isfile(source):
            raise FileNotFoundError(f"story dataset {source} does not exist")
        opened = source = h5py.File(source, "r")
    try:
        grp = source[split]
        decode = decode or _default_image_decoder()
        n = len(grp["text"])
        out = {}
        for i in range(5):
            col = grp[f"image{i}"]
            if len(col) != n:
                raise ValueError(f"{split}/image{i} holds {len(col)} stories, {split}/text {n}")
            out[f"image{i}"] = [decode(b) for b in col]
            if any(im is None or im.ndim != 3 or im.shape[2] != 3 for im in out[f"image{i}"]):
                raise ValueError(f"{split}/image{i}: an entry did not decode to an H x W x 3 image")
        out["text"] = [(t.decode("utf-8") if isinstance(t, (bytes, np.bytes_)) else str(t)).split("|") for t in grp["text"]]
        return out
    finally:
        if opened is not None:
            opened.close()


def pick_story_frames(dataset, index, rng=None, frame_px=128):
    """The five frames of story `index` as the driver cuts them (:257-266): per image slot one of the strip's candidate frames,
    rows [idx * 128, (idx + 1) * 128) with idx = random.randint(0, 4) (rng: anything with .randint(a, b) inclusive — the
    `random` module by default, as there).  -> list of five frame_px x W x 3 uint8 arrays."""
    import random as _random
    rng = rng or _random
    frames = []
    for i in range(5):
        im = dataset[f"image{i}"][index]
        k = im.shape[0] // frame_px
        if k < 1:
            raise ValueError(f"image{i}[{index}] is {im.shape[0]} px high: no {frame_px}-px frame in it")
        idx = rng.randint(0, 4)
        if idx >= k:
            raise ValueError(f"image{i}[{index}] holds {k} candidate frames, the driver draws from 5")
        frames.append(im[idx * frame_px:(idx + 1) * frame_px])
    return frames
