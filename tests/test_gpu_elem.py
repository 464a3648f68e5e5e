"""Small device kernels of the input side (csrc/elem.hip) against their torch statements."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return torch.device("cuda")


def test_text_prep_matches_the_hf_position_ids_and_the_padding_mask(dev):
    """toist_text_prep: RoBERTa's create_position_ids_from_input_ids (cumsum(id != pad) * (id != pad) + pad) and attention_mask != 1
    (/root/reference/models/transformer.py:129-133 through HF RobertaModel) in one launch; ragged captions, a pad id inside a caption."""
    from toist_amd import kernels as k
    torch.manual_seed(0)
    for B, L in ((1, 1), (3, 7), (8, 16), (70, 33)):
        ids = torch.randint(3, 500, (B, L), device=dev)
        lens = torch.randint(1, L + 1, (B,), device=dev)
        att = (torch.arange(L, device=dev)[None, :] < lens[:, None]).long()
        ids = torch.where(att.bool(), ids, torch.ones_like(ids))               # <pad> = 1 behind every caption
        if L > 3:
            ids[0, 1] = 1                                                        # a pad id in the middle (counts as padding for the positions)
        pos, key_pad = k.text_prep(ids, att, 1)
        keep = ids.ne(1).long()
        assert torch.equal(pos, torch.cumsum(keep, dim=1) * keep + 1)
        assert torch.equal(key_pad, att.ne(1).to(torch.uint8))


def test_sine_position_seq_is_the_token_encoding_with_zero_caption_rows(dev):
    """toist_sine_position_seq writes PositionEmbeddingSine's token-major rows straight into the cross-modal sequence layout [B, H*W + L, 256];
    the L caption rows are zero (transformer.py:139); the image rows equal toist_sine_position's bit for bit."""
    from toist_amd.position_encoding import PositionEmbeddingSine
    pe = PositionEmbeddingSine(128, normalize=True)
    torch.manual_seed(1)
    for B, H, W, L in ((2, 5, 7, 3), (8, 20, 20, 16), (1, 1, 1, 1)):
        mask = torch.rand(B, H, W, device=dev) > 0.7
        mask[:, 0, 0] = False
        plain = pe.tokens(mask)
        seq = pe.tokens(mask, tail=L)
        assert seq.shape == (B, H * W + L, 256)
        assert torch.equal(seq[:, :H * W], plain)
        assert not seq[:, H * W:].any()
