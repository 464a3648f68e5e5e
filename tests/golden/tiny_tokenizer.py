"""A REAL Hugging Face fast tokenizer without any downloaded file: byte-level BPE with RoBERTa's special tokens and post-processing (<s> ... </s>, trimmed
offsets), trained once on the fixed corpus below and committed as tests/golden/tiny_roberta_tokenizer.json (vocabulary + merges: data, 400 entries).
`roberta-base`'s own vocabulary cannot be had offline; what the hot path depends on is not the vocabulary but the BatchEncoding mechanics -- padding to the
longest caption, attention masks, and `char_to_token` returning None on the spaces between words, which is what the span lookups of the distillation
losses (/root/reference/models/mdetr.py:112-141, 240-260, 684-711) and their fallbacks act on.  Used by tests/golden/make_golden_distill_tokenizer.py
(fixture from the real reference) and by the tests that replay it."""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "tiny_roberta_tokenizer.json")
CORPUS = ["use the scissors to cut the paper up", "use something to cut the paper up", "step on the wooden stool to reach the shelf", "step on something to reach the shelf",
          "sit comfortably on the armchair", "sit comfortably on something", "pound the nail with the hammer", "pound the nail with something",
          "place the flowers in the vase", "place the flowers in something", "dig a hole with the shovel", "dig a hole with something",
          "open the bottle of beer with the opener", "open the bottle of beer with something", "serve wine in the glass", "serve wine in something"]


def train():
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors, trainers
    tok = Tokenizer(models.BPE(unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=400, special_tokens=["<s>", "<pad>", "</s>", "<unk>", "<mask>"], initial_alphabet=pre_tokenizers.ByteLevel.alphabet(), show_progress=False)
    tok.train_from_iterator(CORPUS * 10, trainer)
    tok.post_processor = processors.RobertaProcessing(sep=("</s>", tok.token_to_id("</s>")), cls=("<s>", tok.token_to_id("<s>")), add_prefix_space=False, trim_offsets=True)
    tok.save(PATH)


def build():
    """-> transformers.PreTrainedTokenizerFast over the committed json (ids 0 / 1 / 2 = <s> / <pad> / </s>, as in roberta-base)"""
    from tokenizers import Tokenizer
    from transformers import PreTrainedTokenizerFast
    return PreTrainedTokenizerFast(tokenizer_object=Tokenizer.from_file(PATH), bos_token="<s>", eos_token="</s>", pad_token="<pad>", unk_token="<unk>", mask_token="<mask>",
                                   cls_token="<s>", sep_token="</s>")


if __name__ == "__main__":
    train()
    t = build()
    e = t(["use the scissors to cut the paper up", "sit comfortably on something"], padding="longest", return_tensors="pt")
    print(e["input_ids"].tolist(), [e.char_to_token(1, c) for c in range(28)])
