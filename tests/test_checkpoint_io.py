"""CPU: checkpoint routing and output-file helpers of the stage-2 driver (rcdms_amd/checkpoint.py) —
a DeepSpeed-layout file written from tiny mirrored modules loads back key for key; prefix routing, rank sharding and the
PNG grid follow stage2_batchtest_rcdms_model.py:58-93,225-243."""
import numpy as np
import pytest
import torch

from rcdms_amd import checkpoint as CK
from rcdms_amd import context, synth
from tests.test_oracle_golden import SEEDS, mirrored, shapes_of


def _modules(seed):
    unet = mirrored("unet_tiny").to_empty(device="cpu")       # built on the meta device: give it storage
    unet.load_state_dict(synth.procedural_state_dict(shapes_of(unet), seed))
    local = context.fine_stack(text_dim=64, vis_dim=32, hidden_dim=64, num_heads=8)
    glob = context.semantic_stack(text_dim=64, vis_dim=24, hidden_dim=64, num_heads=8)
    local.load_state_dict(synth.procedural_state_dict({k: v.shape for k, v in local.state_dict().items()}, seed + 1))
    glob.load_state_dict(synth.procedural_state_dict({k: v.shape for k, v in glob.state_dict().items()}, seed + 2))
    return unet, local, glob


def test_stage2_checkpoint_round_trip(tmp_path):
    src = _modules(SEEDS["unet_tiny"])
    path = CK.stage2_checkpoint_path("exp", 3000, root=str(tmp_path))
    assert path.endswith("exp/3000/mp_rank_00_model_states.pt")
    CK.save_stage2_checkpoint(path, *src)
    raw = torch.load(path, map_location="cpu")
    assert set(raw) == {"module"}
    heads = {k.split(".")[0] for k in raw["module"]}
    assert heads == {"unet", "seen_module", "unseen_module"}
    dst = _modules(99)                                      # different weights, to be overwritten
    other = CK.load_stage2_checkpoint(path, *dst)
    assert other == []
    for a, b in zip(src, dst):
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb)
        assert all(torch.equal(sa[k], sb[k]) for k in sa)


def test_prefix_routing_and_errors(tmp_path):
    sd = {"unet.conv_in.weight": torch.zeros(1), "seen_module.text_fc.fc1.weight": torch.ones(1),
          "unseen_module.vis_fc.fc1.weight": torch.ones(2), "optimizer.step": torch.zeros(1),
          "unet.unet.odd": torch.zeros(3)}
    parts = CK.split_stage2_state(sd)
    assert list(parts["unet"]) == ["conv_in.weight", "unet.odd"]            # head stripped once, not every occurrence
    assert list(parts["local_module"]) == ["text_fc.fc1.weight"] and list(parts["global_module"]) == ["vis_fc.fc1.weight"]
    assert parts["other"] == ["optimizer.step"]
    with pytest.raises(FileNotFoundError):
        CK.load_stage2_checkpoint(str(tmp_path / "nope.pt"), None, None, None)
    with pytest.raises(KeyError, match="seen_module"):
        CK.load_stage2_checkpoint({"module": {"unet.x": torch.zeros(1)}}, None, None, None)
    unet, local, glob = _modules(5)
    bad = {"module": {"unet.conv_in.weight": torch.zeros(1), "seen_module.a": torch.zeros(1), "unseen_module.b": torch.zeros(1)}}
    with pytest.raises(RuntimeError):                        # strict load, as the reference's load_state_dict calls
        CK.load_stage2_checkpoint(bad, unet, local, glob)


def test_split_list_matches_reference_dealing():
    assert CK.split_list(10, 3) == [[0, 1, 2, 3], [4, 5, 6], [7, 8, 9]]
    assert CK.split_list(2, 4) == [[0], [1], [], []]
    for n, m in ((2208, 8), (7, 7), (0, 2)):
        parts = CK.split_list(n, m)
        assert sum(parts, []) == list(range(n)) and max(map(len, parts)) - min(map(len, parts)) <= 1


def test_image_grid_and_frames():
    vid = torch.zeros(1, 3, 5, 4, 6)
    for j in range(5):
        vid[0, :, j] = (j + 1) / 5.0
    vid[0, 0, 2, 1, 3] = 0.999                             # truncation, not rounding: 254
    frames = CK.videos_to_frames(vid)
    assert len(frames) == 1 and len(frames[0]) == 5 and frames[0][0].shape == (4, 6, 3)
    grid = CK.image_grid(frames[0] + frames[0], 2, 5)
    assert grid.size == (30, 8)
    px = np.array(grid)
    assert px[0, 0].tolist() == [51, 51, 51] and px[0, 6].tolist() == [102, 102, 102] and px[4, 24].tolist() == [255] * 3
    assert px[1, 12 + 3].tolist() == [254, 153, 153]
    with pytest.raises(AssertionError):
        CK.image_grid(frames[0], 2, 5)


def test_story_split_reader_on_an_h5_shaped_mapping(tmp_path):
    """The ARLDM test-split read of the driver (stage2_batchtest_rcdms_model.py:41-56,440-453,257-266) on a mapping with the
    file's layout (h5py is not part of this image; an open h5py.File is such a mapping): encoded strips decode to BGR uint8
    arrays exactly as cv2.imdecode(IMREAD_COLOR) returns them, captions split on '|', frames are cut 128 rows at a time."""
    import io
    import numpy as np
    import pytest
    from PIL import Image
    from rcdms_amd.checkpoint import pick_story_frames, read_story_split
    rng = np.random.default_rng(0)

    def strip(seed):
        rgb = np.random.default_rng(seed).integers(0, 256, size=(640, 128, 3), dtype=np.uint8)
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, format="PNG")       # lossless: the decoded pixels are known exactly
        return rgb, np.frombuffer(buf.getvalue(), dtype=np.uint8)

    n = 3
    raw = {i: [strip(10 * i + s) for s in range(n)] for i in range(5)}
    grp = {f"image{i}": [raw[i][s][1] for s in range(n)] for i in range(5)}
    grp["text"] = [("|".join(f"story {s} frame {j}: pororo and loopy" for j in range(5))).encode("utf-8") for s in range(n)]
    data = read_story_split({"test": grp})
    assert sorted(data) == ["image0", "image1", "image2", "image3", "image4", "text"]
    assert all(len(data[k]) == n for k in data)
    for i in range(5):
        for s in range(n):
            assert data[f"image{i}"][s].dtype == np.uint8 and data[f"image{i}"][s].shape == (640, 128, 3)
            assert np.array_equal(data[f"image{i}"][s], raw[i][s][0][:, :, ::-1]), "BGR, as cv2.imdecode returns"
    assert data["text"][1] == [f"story 1 frame {j}: pororo and loopy" for j in range(5)]

    class Fixed:
        def __init__(self, seq):
            self.seq = list(seq)

        def randint(self, a, b):
            assert (a, b) == (0, 4)
            return self.seq.pop(0)

    fr = pick_story_frames(data, 2, rng=Fixed([0, 4, 2, 1, 3]))
    assert [f.shape for f in fr] == [(128, 128, 3)] * 5
    assert np.array_equal(fr[1], data["image1"][2][512:640]) and np.array_equal(fr[2], data["image2"][2][256:384])
    grp["image3"] = grp["image3"][:2]
    with pytest.raises(ValueError, match="image3 holds 2 stories"):
        read_story_split({"test": grp})
    with pytest.raises((ImportError, FileNotFoundError)):
        read_story_split(str(tmp_path / "missing.h5"))
