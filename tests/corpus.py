"""Seeded input generators shared by the CPU and GPU parity tests."""
import random

SPECIALS = [b"<|begin_of_text|>", b"<|end_of_text|>", b"<|start_header_id|>", b"<|end_header_id|>", b"<|eot_id|>",
            b"<|eom_id|>", b"<|python_tag|>", b"<|finetune_right_pad_id|>", b"<|reserved_special_token_0|>",
            b"<|reserved_special_token_9|>", b"<|reserved_special_token_10|>", b"<|reserved_special_token_99|>",
            b"<|reserved_special_token_100|>", b"<|reserved_special_token_239|>", b"<|reserved_special_token_247|>"]
NEAR_MISSES = [b"<|reserved_special_token_248|>", b"<|reserved_special_token_07|>", b"<|eot_id|", b"<|<|eot_id|>",
               b"<|not_special|>", b"<|reserved_special_token_|>", b"<|reserved_special_token_2470|>", b"<|eot_id |>",
               b"<|", b"|>", b"<", b"<|reserved_special_token_24", b"<|begin_of_text|"]

# pieces chosen to hit every branch of the pre-tokenizer rules and the UTF-8 decoder
PIECES = [b"a", b"s", b"S", b"t", b"re", b"RE", b"ve", b"ll", b"m", b"d", b"'", b"'", b" ", b" ", b"  ", b"\t", b"\n", b"\r",
          b"\r\n", b"1", b"12", b"123", b"1234567", b"!", b"?!", b".", b",", b"-", b"_", "é".encode(), "ß".encode(),
          " ".encode(), "　".encode(), " ".encode(), "​".encode(), "\u0085".encode(),
          "²".encode(), "Ⅷ".encode(), "①".encode(), "٣".encode(), "९".encode(), "𝟗".encode(), "中".encode(), "文字".encode(),
          "🦙".encode(), "‍".encode(), "️".encode(), "Ꙋ".encode(), "ſ".encode(), "K".encode(),
          "ã".encode(), "Привет".encode(), "мир".encode(), "αβγ".encode(), "�".encode(),
          b"\xff", b"\x80", b"\xc0\x80", b"\xc2", b"\xe0\x80\x80", b"\xe0\xa0", b"\xed\xa0\x80", b"\xf0\x90\x80",
          b"\xf4\x90\x80\x80", b"\xf5", b"\xe4\xb8", b"\x00", b"\x1c", b"\x7f", b"Hello", b"world", b"The", b" the", b"ing",
          b"don", b"'t", b"'T", b"'Ll", b"'VE", b"x", b"<", b"|", b">", b"<|"]


def random_doc(rng: random.Random, max_pieces=12, special_rate=0.08):
    out = []
    for _ in range(rng.randint(0, max_pieces)):
        r = rng.random()
        if r < special_rate:
            out.append(rng.choice(SPECIALS))
        elif r < special_rate * 1.5:
            out.append(rng.choice(NEAR_MISSES))
        else:
            out.append(rng.choice(PIECES))
    return b"".join(out)


def random_bytes_doc(rng: random.Random, n):
    return bytes(rng.getrandbits(8) for _ in range(n))


def random_word(rng: random.Random, alphabet=b"abcdefghijklmnopqrstuvwxyzABCDEFGH0123456789 '.-_", lo=1, hi=24):
    return bytes(rng.choice(alphabet) for _ in range(rng.randint(lo, hi)))
