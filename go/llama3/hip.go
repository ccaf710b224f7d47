//go:build l3hip

// hip.go -- the cgo shim that puts libl3hip.so (include/l3hip.h) behind the API of package llama3.
//
// Where it goes: copy this file to llama3/hip.go of github.com/agentstation/tokenizer, put include/l3hip.h and
// tokenizer_amd/lib/libl3hip.so under third_party/l3hip/{include,lib}, and build with `-tags l3hip`.  It is written
// against the reference at the commit SURVEY.md names and uses only what that package already exports or defines
// (EncodeOptions, NewTokenError, NewDataError, NewConfigError, Err*, encodeBytes, splitBySpecialTokens,
// optimisticSpecialTokenRegex, beginOfTextToken, endOfTextToken, isSpecialToken, internal/vocabulary, scanner).
// No Go toolchain exists in the image this repository is authored in: the file has NOT been compiled or vetted
// there.  Its behaviour is specified by, and tested through, the C++ twin include/llama3.hpp (tests/test_host_api.py).
//
// What each method stands for (reference file:line):
//   NewHIP                     New                    llama3/tokenizer.go:142-213, options.go:17-74
//   Encode / EncodeBytes       tokenizer.go:217-271   (Encoder, tokenizer.go:48-51)
//   AppendTokens               tokenizer.go:276-330
//   Decode / DecodeBytes       tokenizer.go:333-354   (Decoder, tokenizer.go:55-58)
//   GetSpecialTokenID          tokenizer.go:357-368
//   OptimisticCount            tokenizer.go:373-417
//   VocabSize                  tokenizer.go:420-422
//   PreTokenize                tokenizer.go:490-506   (PreTokenizer, tokenizer.go:89-93)
//   NewScanner / Process / TokenStream   llama3/scanner.go:67-132 over the scanner.Tokenizer seam (scanner/scanner.go:12-15)
//   EncodeBPE                  tokenizer.go:80-84,478-488   (BPE: one pre-token in the byte-level alphabet, L3_ONE_PRETOKEN)
//   EncodeBatch / EncodeBatchPinned      new, additive: the batched hot path

package llama3

/*
#cgo CFLAGS: -I${SRCDIR}/../third_party/l3hip/include
#cgo LDFLAGS: -L${SRCDIR}/../third_party/l3hip/lib -ll3hip -Wl,-rpath,${SRCDIR}/../third_party/l3hip/lib
#include <stdlib.h>
#include "l3hip.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"io"
	"os"
	"runtime"
	"strings"
	"sync"
	"unicode/utf8"
	"unsafe"

	"github.com/agentstation/tokenizer/llama3/internal/vocabulary"
	"github.com/agentstation/tokenizer/llama3/scanner"
)

// HIPTokenizer is the GPU twin of *Tokenizer.  Calls are serialised per tokenizer (the library serialises them per
// context anyway); use one HIPTokenizer per concurrent stream of work.
type HIPTokenizer struct {
	mu    sync.Mutex
	ctx   *C.l3_ctx   // context 0: everything that is not sharded
	multi *C.l3_multi // non-nil with HIPDevices: EncodeBatch shards over several GPUs

	// page-locked staging buffers from l3_host_alloc, grown on demand and reused: with both the text and the ids in
	// page-locked memory l3_encode_batch overlaps upload, kernels and download (about 3.5x the rate of Go-heap buffers)
	pinText, pinIDs         unsafe.Pointer
	pinTextCap, pinIDsCap   int
}

type hipConfig struct {
	devices       []int
	vocabPath     string
	mergesPath    string
	specialTokens []string // nil: the 256 defaults
}

// HIPOption configures NewHIP.
type HIPOption func(*hipConfig) error

// HIPDevice selects the GPU that holds the tables (default 0).
func HIPDevice(device int) HIPOption {
	return func(c *hipConfig) error { c.devices = []int{device}; return nil }
}

// HIPDevices makes EncodeBatch shard every batch, by document, over these GPUs.
func HIPDevices(devices []int) HIPOption {
	return func(c *hipConfig) error {
		if len(devices) == 0 {
			return NewConfigError("devices", devices, ErrInvalidToken)
		}
		c.devices = append([]int(nil), devices...)
		return nil
	}
}

// HIPDataFiles is WithDataFiles (options.go:65-74).
func HIPDataFiles(vocabPath, mergesPath string) HIPOption {
	return func(c *hipConfig) error { c.vocabPath, c.mergesPath = vocabPath, mergesPath; return nil }
}

// HIPSpecialTokens is WithSpecialTokens (options.go:17-37), with the same validation.
func HIPSpecialTokens(tokens []string) HIPOption {
	return func(c *hipConfig) error {
		probe := &config{}
		if err := WithSpecialTokens(tokens)(probe); err != nil {
			return err
		}
		c.specialTokens = tokens
		return nil
	}
}

// NewHIP mirrors New (tokenizer.go:142-213): same data (embedded, or the two files), same special-token rule, tables
// built once and uploaded to the GPU(s).
func NewHIP(opts ...HIPOption) (*HIPTokenizer, error) {
	cfg := &hipConfig{devices: []int{0}}
	for _, o := range opts {
		if err := o(cfg); err != nil {
			return nil, err
		}
	}
	vocab, merges := vocabulary.EmbeddedVocabulary, vocabulary.EmbeddedMergeRules // internal/vocabulary/embed.go:13-20
	if cfg.vocabPath != "" {
		v, err := os.ReadFile(cfg.vocabPath)
		if err != nil {
			return nil, NewDataError("load vocabulary", cfg.vocabPath, ErrDataNotFound)
		}
		m, err := os.ReadFile(cfg.mergesPath)
		if err != nil {
			return nil, NewDataError("load merges", cfg.mergesPath, ErrDataNotFound)
		}
		vocab, merges = string(v), string(m)
	}

	// the special-token list as a C array of C strings (nil: the defaults)
	var names **C.char
	nNames := 0
	if cfg.specialTokens != nil {
		nNames = len(cfg.specialTokens)
		arr := C.malloc(C.size_t(nNames+1) * C.size_t(unsafe.Sizeof(uintptr(0))))
		defer C.free(arr)
		slots := unsafe.Slice((**C.char)(arr), nNames+1)
		for i, s := range cfg.specialTokens {
			slots[i] = C.CString(s)
			defer C.free(unsafe.Pointer(slots[i]))
		}
		slots[nNames] = nil
		names = (**C.char)(arr)
	}

	t := &HIPTokenizer{}
	vp, mp := unsafe.Pointer(unsafe.StringData(vocab)), unsafe.Pointer(unsafe.StringData(merges))
	var rc C.int
	if len(cfg.devices) > 1 {
		devs := make([]C.int, len(cfg.devices))
		for i, d := range cfg.devices {
			devs[i] = C.int(d)
		}
		rc = C.l3_multi_create(vp, C.size_t(len(vocab)), mp, C.size_t(len(merges)), names, C.size_t(nNames),
			(*C.int)(unsafe.SliceData(devs)), C.int(len(devs)), &t.multi)
		if rc == C.L3_OK {
			t.ctx = C.l3_multi_context(t.multi, 0)
		}
	} else {
		rc = C.l3_create_ex(vp, C.size_t(len(vocab)), mp, C.size_t(len(merges)), names, C.size_t(nNames),
			C.int(cfg.devices[0]), &t.ctx)
	}
	if rc != C.L3_OK {
		return nil, NewDataError("l3_create", C.GoString(C.l3_last_error(nil)), ErrDataNotFound)
	}
	runtime.SetFinalizer(t, func(t *HIPTokenizer) { t.Close() })
	return t, nil
}

// Close releases the GPU tables and the page-locked buffers.  The tokenizer must not be used afterwards.
func (t *HIPTokenizer) Close() {
	t.mu.Lock()
	defer t.mu.Unlock()
	if t.ctx == nil {
		return
	}
	if t.pinText != nil {
		C.l3_host_free(t.ctx, t.pinText)
	}
	if t.pinIDs != nil {
		C.l3_host_free(t.ctx, t.pinIDs)
	}
	t.pinText, t.pinIDs, t.pinTextCap, t.pinIDsCap = nil, nil, 0, 0
	if t.multi != nil {
		C.l3_multi_destroy(t.multi) // owns every context
	} else {
		C.l3_destroy(t.ctx)
	}
	t.ctx, t.multi = nil, nil
	runtime.SetFinalizer(t, nil)
}

func (t *HIPTokenizer) lastError() error {
	if t.multi != nil {
		return errors.New(C.GoString(C.l3_multi_last_error(t.multi)))
	}
	return errors.New(C.GoString(C.l3_last_error(t.ctx)))
}

func hipFlags(o *EncodeOptions) C.uint32_t { // nil => BOS and EOS (tokenizer.go:218-220)
	if o == nil {
		return C.L3_BOS | C.L3_EOS
	}
	var f C.uint32_t
	if o.BOS {
		f |= C.L3_BOS
	}
	if o.EOS {
		f |= C.L3_EOS
	}
	return f
}

// encodeRaw is one l3_encode_batch / l3_multi_encode_batch call with the capacity protocol of the header: on
// L3_E_OUTPUT_CAP *n holds the size needed and grow() supplies a buffer of at least that size.
func (t *HIPTokenizer) encodeRaw(text unsafe.Pointer, docOff []uint64, flags C.uint32_t, ids unsafe.Pointer, idsCap int,
	off []uint64, grow func(need int) (unsafe.Pointer, int, error)) (int, unsafe.Pointer, error) {
	nDocs := C.uint64_t(len(docOff) - 1)
	var n C.uint64_t
	for {
		var rc C.int
		if t.multi != nil {
			rc = C.l3_multi_encode_batch(t.multi, (*C.uint8_t)(text), (*C.uint64_t)(unsafe.SliceData(docOff)), nDocs, flags,
				(*C.int32_t)(ids), C.uint64_t(idsCap), (*C.uint64_t)(unsafe.SliceData(off)), &n)
		} else {
			rc = C.l3_encode_batch(t.ctx, (*C.uint8_t)(text), (*C.uint64_t)(unsafe.SliceData(docOff)), nDocs, flags,
				(*C.int32_t)(ids), C.uint64_t(idsCap), (*C.uint64_t)(unsafe.SliceData(off)), &n)
		}
		if rc == C.L3_E_OUTPUT_CAP {
			var err error
			if ids, idsCap, err = grow(int(n)); err != nil {
				return 0, nil, err
			}
			continue
		}
		if rc != C.L3_OK {
			return 0, nil, t.lastError()
		}
		return int(n), ids, nil
	}
}

// EncodeBatch is the new, additive call: document i is text[docOff[i]:docOff[i+1]] (docOff[0] == 0, ascending, the last
// entry len(text)); its ids are ids[idOff[i]:idOff[i+1]].  Results are bit-identical to calling Encode per document.
// Buffers are ordinary Go slices; for the host-fed throughput figure use EncodeBatchPinned.
func (t *HIPTokenizer) EncodeBatch(text []byte, docOff []uint64, o *EncodeOptions) (ids []int32, idOff []uint64, err error) {
	if len(docOff) == 0 || docOff[0] != 0 || docOff[len(docOff)-1] != uint64(len(text)) {
		return nil, nil, NewConfigError("docOff", len(docOff), ErrInvalidToken)
	}
	t.mu.Lock()
	defer t.mu.Unlock()
	nDocs := len(docOff) - 1
	out := make([]int32, len(text)/3+2*nDocs+64)
	idOff = make([]uint64, nDocs+1)
	grow := func(need int) (unsafe.Pointer, int, error) {
		out = make([]int32, need)
		return unsafe.Pointer(unsafe.SliceData(out)), need, nil
	}
	n, _, err := t.encodeRaw(unsafe.Pointer(unsafe.SliceData(text)), docOff, hipFlags(o),
		unsafe.Pointer(unsafe.SliceData(out)), len(out), idOff, grow)
	runtime.KeepAlive(text)
	if err != nil {
		return nil, nil, err
	}
	return out[:n], idOff, nil
}

// PinnedText returns a page-locked buffer of at least n bytes that stays valid until the next PinnedText call or
// Close: fill it with the batch text and pass it to EncodeBatchPinned.
func (t *HIPTokenizer) PinnedText(n int) ([]byte, error) {
	t.mu.Lock()
	defer t.mu.Unlock()
	if n > t.pinTextCap {
		if t.pinText != nil {
			C.l3_host_free(t.ctx, t.pinText)
			t.pinText, t.pinTextCap = nil, 0
		}
		want := n + n/8 + 4096
		if rc := C.l3_host_alloc(t.ctx, C.size_t(want), &t.pinText); rc != C.L3_OK {
			return nil, t.lastError()
		}
		t.pinTextCap = want
	}
	return unsafe.Slice((*byte)(t.pinText), t.pinTextCap)[:n], nil
}

func (t *HIPTokenizer) growPinnedIDs(need int) (unsafe.Pointer, int, error) {
	if need > t.pinIDsCap {
		if t.pinIDs != nil {
			C.l3_host_free(t.ctx, t.pinIDs)
			t.pinIDs, t.pinIDsCap = nil, 0
		}
		want := need + need/8 + 1024
		if rc := C.l3_host_alloc(t.ctx, C.size_t(want)*4, &t.pinIDs); rc != C.L3_OK {
			return nil, 0, t.lastError()
		}
		t.pinIDsCap = want
	}
	return t.pinIDs, t.pinIDsCap, nil
}

// EncodeBatchPinned is EncodeBatch for text that lies in the buffer PinnedText returned.  The ids come back in a
// page-locked buffer of the tokenizer's: the slice is valid until the next EncodeBatchPinned call or Close.
func (t *HIPTokenizer) EncodeBatchPinned(text []byte, docOff []uint64, o *EncodeOptions) (ids []int32, idOff []uint64, err error) {
	if len(docOff) == 0 || docOff[0] != 0 || docOff[len(docOff)-1] != uint64(len(text)) {
		return nil, nil, NewConfigError("docOff", len(docOff), ErrInvalidToken)
	}
	t.mu.Lock()
	defer t.mu.Unlock()
	nDocs := len(docOff) - 1
	buf, bufCap, err := t.growPinnedIDs(len(text)/3 + 2*nDocs + 1024)
	if err != nil {
		return nil, nil, err
	}
	idOff = make([]uint64, nDocs+1)
	n, buf, err := t.encodeRaw(unsafe.Pointer(unsafe.SliceData(text)), docOff, hipFlags(o), buf, bufCap, idOff, t.growPinnedIDs)
	if err != nil {
		return nil, nil, err
	}
	return unsafe.Slice((*int32)(buf), n), idOff, nil
}

// Encode keeps the reference signature (tokenizer.go:217): a batch of one document.  The reference's Encode cannot fail;
// a device failure yields nil here, callers that care use EncodeBatch.
func (t *HIPTokenizer) Encode(text string, o *EncodeOptions) []int {
	return t.AppendTokens(nil, text, o)
}

// EncodeBytes is tokenizer.go:269-271.
func (t *HIPTokenizer) EncodeBytes(data []byte, o *EncodeOptions) []int {
	ids, _, err := t.EncodeBatch(data, []uint64{0, uint64(len(data))}, o)
	if err != nil {
		return nil
	}
	r := make([]int, len(ids))
	for i, v := range ids {
		r[i] = int(v)
	}
	return r
}

// AppendTokens is tokenizer.go:276-330: Encode(text) appended to dst.
func (t *HIPTokenizer) AppendTokens(dst []int, text string, o *EncodeOptions) []int {
	ids, _, err := t.EncodeBatch(unsafe.Slice(unsafe.StringData(text), len(text)), []uint64{0, uint64(len(text))}, o)
	if err != nil {
		return dst
	}
	for _, v := range ids {
		dst = append(dst, int(v))
	}
	return dst
}

// Decode is tokenizer.go:333-335.
func (t *HIPTokenizer) Decode(tokenIDs []int) string { return string(t.DecodeBytes(tokenIDs)) }

// DecodeBytes is tokenizer.go:339-354: ids outside [0, VocabSize) are skipped.
func (t *HIPTokenizer) DecodeBytes(tokenIDs []int) []byte {
	t.mu.Lock()
	defer t.mu.Unlock()
	ids := make([]int32, len(tokenIDs))
	for i, v := range tokenIDs {
		if v < 0 || v > 0x7fffffff {
			ids[i] = -1 // skipped by the kernel like every id outside the vocabulary
		} else {
			ids[i] = int32(v)
		}
	}
	off := []uint64{0, uint64(len(ids))}
	outOff := make([]uint64, 2)
	out := make([]byte, len(ids)*8+64)
	var n C.uint64_t
	for {
		rc := C.l3_decode_batch(t.ctx, (*C.int32_t)(unsafe.SliceData(ids)), (*C.uint64_t)(unsafe.SliceData(off)), 1,
			(*C.uint8_t)(unsafe.SliceData(out)), C.uint64_t(len(out)), (*C.uint64_t)(unsafe.SliceData(outOff)), &n)
		if rc == C.L3_E_OUTPUT_CAP {
			out = make([]byte, int(n))
			continue
		}
		if rc != C.L3_OK {
			return nil
		}
		return out[:n]
	}
}

// GetSpecialTokenID is tokenizer.go:357-368, error values included.
func (t *HIPTokenizer) GetSpecialTokenID(token string) (int, error) {
	if !isSpecialToken(token) {
		return 0, NewTokenError("validate special token", token, ErrInvalidToken)
	}
	var id C.int32_t
	rc := C.l3_special_token_id(t.ctx, (*C.char)(unsafe.Pointer(unsafe.StringData(token))), C.size_t(len(token)), &id)
	if rc != C.L3_OK {
		return 0, NewTokenError("get special token ID", token, ErrTokenNotFound)
	}
	return int(id), nil
}

// VocabSize is tokenizer.go:420-422.
func (t *HIPTokenizer) VocabSize() int { return int(C.l3_vocab_size(t.ctx)) }

// OptimisticCount is tokenizer.go:373-417: anything shaped <|[a-zA-Z0-9_]+|> counts as ONE token, BOS and EOS are
// counted when the tokenizer has them, and the text between is encoded as plain text -- all pieces in one batch call,
// a document per piece, with L3_NO_SPECIAL.  -1 on a device failure.
func (t *HIPTokenizer) OptimisticCount(text string) int {
	count := 0
	if _, err := t.GetSpecialTokenID(beginOfTextToken); err == nil {
		count++
	}
	var plain []byte
	docOff := []uint64{0}
	for _, piece := range splitBySpecialTokens(text, optimisticSpecialTokenRegex) {
		if optimisticSpecialTokenRegex.MatchString(piece) {
			count++
			continue
		}
		if piece != "" {
			plain = append(plain, piece...)
			docOff = append(docOff, uint64(len(plain)))
		}
	}
	if len(docOff) > 1 {
		t.mu.Lock()
		out := make([]int32, len(plain)+64)
		idOff := make([]uint64, len(docOff))
		n, _, err := t.encodeRaw(unsafe.Pointer(unsafe.SliceData(plain)), docOff, C.L3_NO_SPECIAL,
			unsafe.Pointer(unsafe.SliceData(out)), len(out), idOff, func(need int) (unsafe.Pointer, int, error) {
				out = make([]int32, need)
				return unsafe.Pointer(unsafe.SliceData(out)), need, nil
			})
		t.mu.Unlock()
		runtime.KeepAlive(plain)
		if err != nil {
			return -1
		}
		count += n
	}
	if _, err := t.GetSpecialTokenID(endOfTextToken); err == nil {
		count++
	}
	return count
}

// PreTokenize is tokenizer.go:490-506: the pieces of pretokenizer.Tokenize(text) -- which knows no special tokens --
// each through encodeBytes.  The GPU marks the piece starts (l3_pretokenize_batch with L3_NO_SPECIAL); as in the
// reference the text passes through []rune first, so every byte outside a valid UTF-8 sequence has become U+FFFD
// before the byte-level mapping.
func (t *HIPTokenizer) PreTokenize(text string) []string {
	if text == "" {
		return []string{}
	}
	t.mu.Lock()
	kinds := make([]byte, len(text))
	off := []uint64{0, uint64(len(text))}
	rc := C.l3_pretokenize_batch(t.ctx, (*C.uint8_t)(unsafe.Pointer(unsafe.StringData(text))), (*C.uint64_t)(unsafe.SliceData(off)), 1,
		C.L3_NO_SPECIAL, (*C.uint8_t)(unsafe.SliceData(kinds)))
	t.mu.Unlock()
	if rc != C.L3_OK {
		return nil
	}
	var parts []string
	var cur strings.Builder
	for i := 0; i < len(text); {
		if i > 0 && kinds[i] != 0 {
			parts = append(parts, encodeBytes([]byte(cur.String())))
			cur.Reset()
		}
		r, w := utf8.DecodeRuneInString(text[i:])
		cur.WriteRune(r) // utf8.RuneError for an invalid byte, w == 1: what string -> []rune -> string makes of it
		i += w
	}
	parts = append(parts, encodeBytes([]byte(cur.String())))
	return parts
}

// EncodeBPE is the BPE interface (tokenizer.go:80-84,478-488 -> bpe.go:18-71): the tokens of ONE pre-token given in the
// byte-level alphabet (what PreTokenize returns).  The reference's lookup table holds the special tokens, so the whole
// string is looked up among them first; then the alphabet is inverted (runes outside it are skipped, bpe.go:84-92) and the
// bytes go to the GPU as a document that is one pre-token (L3_ONE_PRETOKEN).  nil on a device failure.
func (t *HIPTokenizer) EncodeBPE(pretoken string) []int {
	if id, err := t.GetSpecialTokenID(pretoken); err == nil {
		return []int{id}
	}
	raw := make([]byte, 0, len(pretoken))
	for _, r := range pretoken {
		if b, ok := byteLevelInverse[r]; ok {
			raw = append(raw, b)
		}
	}
	if len(raw) == 0 {
		return []int{}
	}
	t.mu.Lock()
	out := make([]int32, len(raw)+8)
	idOff := make([]uint64, 2)
	n, _, err := t.encodeRaw(unsafe.Pointer(unsafe.SliceData(raw)), []uint64{0, uint64(len(raw))}, C.L3_ONE_PRETOKEN,
		unsafe.Pointer(unsafe.SliceData(out)), len(out), idOff, func(need int) (unsafe.Pointer, int, error) {
			out = make([]int32, need)
			return unsafe.Pointer(unsafe.SliceData(out)), need, nil
		})
	t.mu.Unlock()
	runtime.KeepAlive(raw)
	if err != nil {
		return nil
	}
	ids := make([]int, n)
	for i := range ids {
		ids[i] = int(out[i])
	}
	return ids
}

// byteLevelInverse: rune of the byte-level alphabet (internal/encoding/encoding.go:26-79) -> its byte.
var byteLevelInverse = func() map[rune]byte {
	m := make(map[rune]byte, 256)
	n := 0
	for b := 0; b < 256; b++ {
		if (b >= 0x21 && b <= 0x7E) || (b >= 0xA1 && b <= 0xAC) || b >= 0xAE {
			m[rune(b)] = byte(b)
		} else {
			m[rune(256+n)] = byte(b)
			n++
		}
	}
	return m
}()

// hipAdapter fills the scanner.Tokenizer seam (scanner/scanner.go:12-15), as tokenizerAdapter does for *Tokenizer
// (llama3/scanner.go:52-62).
type hipAdapter struct{ *HIPTokenizer }

func (a *hipAdapter) Encode(text string, opts *scanner.EncodeOptions) []int {
	return a.HIPTokenizer.Encode(text, &EncodeOptions{BOS: opts.BOS, EOS: opts.EOS})
}

// NewScanner is llama3/scanner.go:67-72.
func (t *HIPTokenizer) NewScanner(r io.Reader, opts ...ScannerOption) Scanner {
	if len(opts) == 0 {
		return scanner.New(&hipAdapter{t}, r)
	}
	return scanner.NewWithOptions(&hipAdapter{t}, r, opts...)
}

// Process is llama3/scanner.go:77-102: every token as 4 little-endian bytes.
func (t *HIPTokenizer) Process(r io.Reader, w io.Writer) (int64, error) {
	scan := t.NewScanner(r)
	var count int64
	var buf [4]byte
	for scan.Scan() {
		token := scan.Token()
		buf[0], buf[1], buf[2], buf[3] = byte(token), byte(token>>8), byte(token>>16), byte(token>>24)
		if _, err := w.Write(buf[:]); err != nil {
			return count, fmt.Errorf("write token: %w", err)
		}
		count++
	}
	return count, scan.Err()
}

// TokenStream is llama3/scanner.go:107-132.
func (t *HIPTokenizer) TokenStream(r io.Reader) (<-chan int, <-chan error) {
	tokens := make(chan int, 100)
	errc := make(chan error, 1)
	go func() {
		defer close(tokens)
		defer close(errc)
		scan := t.NewScanner(r)
		for scan.Scan() {
			tokens <- scan.Token()
		}
		if err := scan.Err(); err != nil {
			errc <- err
		}
	}()
	return tokens, errc
}

// EncodeStreams is SURVEY 8 row f-1: the scanner's chunks of MANY streams encoded by ONE batch call.  The chunks are
// taken with a recording scanner.Tokenizer (which encodes nothing), then encoded together; per stream the result equals
// what a Scanner over that stream yields.  include/llama3.hpp (EncodeStreams) is the tested twin.
func (t *HIPTokenizer) EncodeStreams(readers []io.Reader, o *EncodeOptions, opts ...ScannerOption) ([][]int, error) {
	if o == nil {
		o = &EncodeOptions{} // scanner.NewWithOptions defaults to no BOS / EOS (scanner/scanner.go:98)
	}
	var text []byte
	docOff := []uint64{0}
	first := make([]int, len(readers)+1)
	for i, r := range readers {
		first[i] = len(docOff) - 1
		rec := &chunkRecorder{}
		sc := scanner.NewWithOptions(rec, r, opts...)
		for sc.Scan() {
		}
		if err := sc.Err(); err != nil {
			return nil, err
		}
		for _, c := range rec.chunks {
			text = append(text, c...)
			docOff = append(docOff, uint64(len(text)))
		}
	}
	first[len(readers)] = len(docOff) - 1
	ids, idOff, err := t.EncodeBatch(text, docOff, &EncodeOptions{})
	if err != nil {
		return nil, err
	}
	bos, bosErr := t.GetSpecialTokenID(beginOfTextToken)
	eos, eosErr := t.GetSpecialTokenID(endOfTextToken)
	out := make([][]int, len(readers))
	for i := range readers {
		var v []int
		if o.BOS && bosErr == nil {
			v = append(v, bos)
		}
		for _, id := range ids[idOff[first[i]]:idOff[first[i+1]]] {
			v = append(v, int(id))
		}
		if o.EOS && eosErr == nil {
			v = append(v, eos)
		}
		out[i] = v
	}
	return out, nil
}

// chunkRecorder is a scanner.Tokenizer that keeps the text of every chunk and yields one placeholder token per chunk,
// so that the scanner keeps going exactly as it would with a real tokenizer.
type chunkRecorder struct{ chunks []string }

func (c *chunkRecorder) Encode(text string, _ *scanner.EncodeOptions) []int {
	c.chunks = append(c.chunks, text)
	return []int{0}
}
func (c *chunkRecorder) GetSpecialTokenID(string) (int, error) { return 0, ErrTokenNotFound }

// HIPTokenizer fills the same interfaces as Tokenizer.
var (
	_ Encoder           = (*HIPTokenizer)(nil)
	_ BPE               = (*HIPTokenizer)(nil)
	_ Decoder           = (*HIPTokenizer)(nil)
	_ PreTokenizer      = (*HIPTokenizer)(nil)
	_ scanner.Tokenizer = (*hipAdapter)(nil)
)
