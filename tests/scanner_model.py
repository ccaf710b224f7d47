"""Python restatement of the reference scanner's read/accumulate rules (llama3/scanner/scanner.go:124-181,
281-311, 337-478): which chunks of a stream reach Encode.  Test infrastructure: the checker for the C++
Chunker in include/llama3.hpp.  The reader model follows scanner_test.go:458-489 (controlledReader: the given
read sizes, then as much as the caller allows, EOF on the read after the last byte)."""


def _is_valid_utf8_ending(b):  # scanner.go:370-403
    if not b:
        return True
    last = b[-1]
    if last < 0x80:
        return True
    if last & 0xC0 == 0x80:
        return False
    # a lead byte at the very end expects more bytes; F8..FF are no lead bytes and count as complete
    return not (last & 0xE0 == 0xC0 or last & 0xF0 == 0xE0 or last & 0xF8 == 0xF0)


def _find_last_complete_utf8(b):  # scanner.go:406-438
    n = len(b)
    i = n - 1
    while i >= 0 and i >= n - 4:
        c = b[i]
        if c < 0x80:
            return i + 1
        if c & 0xC0 != 0x80:
            seq = 2 if c & 0xE0 == 0xC0 else 3 if c & 0xF0 == 0xE0 else 4 if c & 0xF8 == 0xF0 else 0
            return i + seq if i + seq <= n else i
        i -= 1
    return n


def _find_utf8_boundary(d, max_bytes):  # scanner.go:443-478
    n = len(d)
    if max_bytes >= n:
        return n
    i = max_bytes
    while i > 0 and i > max_bytes - 4:
        if i < n:
            b = d[i]
            if b < 0x80 or b & 0xC0 != 0x80:
                return i
        i -= 1
    if max_bytes < n and d[max_bytes] & 0xC0 == 0x80:
        i = max_bytes - 1
        while i >= 0 and i >= max_bytes - 4:
            if d[i] & 0xC0 != 0x80:
                return i
            i -= 1
    return max_bytes


def chunks(data: bytes, read_sizes=(), buf_size=4096, max_buffer=1 << 20):
    """-> (list of (chunk, is_last)) ; is_last = the scanner knew the input was at EOF when it encoded the chunk.
    A trailing (b"", True) entry stands for the empty read at EOF (BOS/EOS-only tokens, scanner.go:184-205)."""
    out = []
    pos = 0
    k = 0
    text = b""
    pending = b""
    done = False
    while not (done and not text):
        while True:
            # readMoreData
            if pos >= len(data):
                n, eof = 0, True
            else:
                want = read_sizes[k] if k < len(read_sizes) else buf_size
                k += 1
                want = min(want, buf_size, len(data) - pos)
                n, eof = want, False
            if n > 0:
                to_write = pending + data[pos:pos + n]
                pending = b""
                pos += n
                if len(text) + len(to_write) > max_buffer:  # handleBufferLimit
                    max_write = max_buffer - len(text)
                    if 0 < max_write < len(to_write):
                        up = _find_utf8_boundary(to_write, max_write)
                        if up < len(to_write):
                            pending = to_write[up:]
                            to_write = to_write[:up]
                text += to_write
            if eof:
                done = True
                text += pending
                pending = b""
            if len(text) >= max_buffer:  # handleMaxBufferReached
                if len(text) == max_buffer and not pending and text and not _is_valid_utf8_ending(text):
                    s = _find_last_complete_utf8(text)
                    if s < len(text):
                        pending = text[s:]
                        text = text[:s]
                break
            if done:
                break
            last = text[-1] if text else None  # hasTokenizationBoundary, scanner.go:337-367
            if last is not None:
                if last in (0x20, 0x0A, 0x09, 0x0D):
                    break
                if last & 0xC0 != 0x80 and len(text) > buf_size // 2:
                    break
        out.append((text, done))
        text = b""
        if done:
            break
    return out
