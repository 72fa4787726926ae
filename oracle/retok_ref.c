/*
 * retok_ref.c — plain-C oracle of the retokenizer (TEST INFRASTRUCTURE, see
 * oracle/__init__.py; never linked into libzett_hip.so).
 *
 * Restates get_surface_form_matrix (reference zett/utils.py:651-689) and the
 * third-party algorithm it calls per token (zett/utils.py:681): HF `tokenizers`
 * 0.22.2 `BPE::tokenize` (merge_word + Word::merge_all), `Unigram::tokenize`
 * (encode_optimized Viterbi + piece->id with byte fallback) and `WordPiece::tokenize`
 * (greedy longest match, continuing pieces looked up with their prefix).  Works in RAW BYTE
 * space: every byte-level character is one byte (CHARS_TO_BYTES, zett/utils.py:351-609).
 *
 * Deliberately simple data structures (sorted arrays + bsearch, O(n^2) loops): it has
 * to be obviously right, not fast.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const uint8_t* p;
    int len;
    int id;
    double score;
    int order;      /* position in the vocabulary listing: the LAST duplicate wins */
} piece_t;

typedef struct {
    int a, b, rank, new_id;
} merge_t;

typedef struct {
    int kind;                 /* 0 BPE, 1 Unigram, 2 WordPiece */
    int n_pieces;
    piece_t* pieces;          /* sorted by bytes; duplicates collapsed to the last one */
    uint8_t* blob;
    int max_piece_len;
    int n_merges;
    merge_t* merges;          /* sorted by (a,b); duplicates collapsed to the last one */
    int unk_id, fuse_unk, byte_fallback, ignore_merges;
    int bf_ids[256];
    double min_score;
    int n_special;
    piece_t* specials;
    uint8_t* sblob;
    int single_id[256];       /* id of the one-byte piece, -1 = absent */
    /* WordPiece: vocabulary entries that start with continuing_subword_prefix, prefix stripped (what a lookup at
       start > 0 can match); `pieces` holds every entry as listed (what a lookup at start == 0 can match) */
    int n_cont;
    piece_t* cont;
    uint8_t* cblob;
    int max_cont_len;
    int max_chars;            /* max_input_chars_per_word */
} model_t;

/* UTF-8 bytes of the printable character that stands for raw byte b (GPT-2 table):
 * bytes 33..126, 161..172, 174..255 map to themselves, the others to U+0100+n. */
static int char_utf8(int b, uint8_t out[2]) {
    int cp;
    if ((b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174 && b <= 255)) {
        cp = b;
    } else {
        int n = 0, i;
        for (i = 0; i < b; ++i)
            if (!((i >= 33 && i <= 126) || (i >= 161 && i <= 172) || (i >= 174 && i <= 255))) ++n;
        cp = 256 + n;
    }
    if (cp < 0x80) { out[0] = (uint8_t)cp; return 1; }
    out[0] = (uint8_t)(0xC0 | (cp >> 6));
    out[1] = (uint8_t)(0x80 | (cp & 0x3F));
    return 2;
}

static int cmp_bytes(const uint8_t* a, int la, const uint8_t* b, int lb) {
    int n = la < lb ? la : lb;
    int c = memcmp(a, b, (size_t)n);
    if (c) return c;
    return la - lb;
}

static int cmp_piece(const void* x, const void* y) {
    const piece_t* a = (const piece_t*)x;
    const piece_t* b = (const piece_t*)y;
    int c = cmp_bytes(a->p, a->len, b->p, b->len);
    if (c) return c;
    return a->order - b->order;
}

static int cmp_merge(const void* x, const void* y) {
    const merge_t* a = (const merge_t*)x;
    const merge_t* b = (const merge_t*)y;
    if (a->a != b->a) return a->a < b->a ? -1 : 1;
    if (a->b != b->b) return a->b < b->b ? -1 : 1;
    return a->rank - b->rank;
}

static const piece_t* find_piece(const piece_t* arr, int n, const uint8_t* p, int len) {
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        int c = cmp_bytes(arr[mid].p, arr[mid].len, p, len);
        if (c == 0) return &arr[mid];
        if (c < 0) lo = mid + 1; else hi = mid - 1;
    }
    return NULL;
}

static const merge_t* find_merge(const model_t* m, int a, int b) {
    int lo = 0, hi = m->n_merges - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        const merge_t* e = &m->merges[mid];
        if (e->a == a && e->b == b) return e;
        if (e->a < a || (e->a == a && e->b < b)) lo = mid + 1; else hi = mid - 1;
    }
    return NULL;
}

static int dedup_pieces(piece_t* arr, int n) {   /* keep the last listed duplicate */
    int i, w = 0;
    for (i = 0; i < n; ++i) {
        if (i + 1 < n && cmp_bytes(arr[i].p, arr[i].len, arr[i + 1].p, arr[i + 1].len) == 0) continue;
        arr[w++] = arr[i];
    }
    return w;
}

void* retok_ref_new(int kind, int n_pieces, const uint8_t* piece_bytes, const int32_t* piece_offsets,
                    const int32_t* piece_ids, const double* piece_scores, int n_merges, const int32_t* merges,
                    int unk_id, int fuse_unk, int byte_fallback, const int32_t* bf_ids, int ignore_merges,
                    double min_score, int n_special, const uint8_t* special_bytes, const int32_t* special_offsets,
                    const int32_t* special_ids) {
    model_t* m = (model_t*)calloc(1, sizeof(model_t));
    int i, total;
    m->kind = kind;
    m->unk_id = unk_id; m->fuse_unk = fuse_unk; m->byte_fallback = byte_fallback; m->ignore_merges = ignore_merges;
    m->min_score = min_score;
    for (i = 0; i < 256; ++i) { m->bf_ids[i] = bf_ids ? bf_ids[i] : -1; m->single_id[i] = -1; }
    total = n_pieces ? piece_offsets[n_pieces] : 0;
    m->blob = (uint8_t*)malloc((size_t)total + 1);
    if (total) memcpy(m->blob, piece_bytes, (size_t)total);
    m->pieces = (piece_t*)malloc(sizeof(piece_t) * (size_t)(n_pieces + 1));
    for (i = 0; i < n_pieces; ++i) {
        piece_t* p = &m->pieces[i];
        p->p = m->blob + piece_offsets[i];
        p->len = piece_offsets[i + 1] - piece_offsets[i];
        p->id = piece_ids[i];
        p->score = piece_scores ? piece_scores[i] : 0.0;
        p->order = i;
        if (p->len > m->max_piece_len) m->max_piece_len = p->len;
    }
    qsort(m->pieces, (size_t)n_pieces, sizeof(piece_t), cmp_piece);
    m->n_pieces = dedup_pieces(m->pieces, n_pieces);
    for (i = 0; i < m->n_pieces; ++i)
        if (m->pieces[i].len == 1) m->single_id[m->pieces[i].p[0]] = m->pieces[i].id;
    m->merges = (merge_t*)malloc(sizeof(merge_t) * (size_t)(n_merges + 1));
    for (i = 0; i < n_merges; ++i) {
        m->merges[i].a = merges[3 * i]; m->merges[i].b = merges[3 * i + 1];
        m->merges[i].new_id = merges[3 * i + 2]; m->merges[i].rank = i;
    }
    qsort(m->merges, (size_t)n_merges, sizeof(merge_t), cmp_merge);
    {   /* a pair listed twice: the later (higher-rank) entry wins, as in a HashMap collect */
        int w = 0;
        for (i = 0; i < n_merges; ++i) {
            if (i + 1 < n_merges && m->merges[i].a == m->merges[i + 1].a && m->merges[i].b == m->merges[i + 1].b) continue;
            m->merges[w++] = m->merges[i];
        }
        m->n_merges = w;
    }
    total = n_special ? special_offsets[n_special] : 0;
    m->sblob = (uint8_t*)malloc((size_t)total + 1);
    if (total) memcpy(m->sblob, special_bytes, (size_t)total);
    m->specials = (piece_t*)malloc(sizeof(piece_t) * (size_t)(n_special + 1));
    for (i = 0; i < n_special; ++i) {
        m->specials[i].p = m->sblob + special_offsets[i];
        m->specials[i].len = special_offsets[i + 1] - special_offsets[i];
        m->specials[i].id = special_ids[i];
        m->specials[i].order = i;
        m->specials[i].score = 0.0;
    }
    qsort(m->specials, (size_t)n_special, sizeof(piece_t), cmp_piece);
    /* duplicates among specials: dict semantics of the Python side do not matter here,
       convert_tokens_to_ids is a function of the string; keep the last */
    m->n_special = dedup_pieces(m->specials, n_special);
    return m;
}

void retok_ref_set_wordpiece(void* h, int n_cont, const uint8_t* cont_bytes, const int32_t* cont_offsets,
                             const int32_t* cont_ids, int max_chars) {
    model_t* m = (model_t*)h;
    int i, total = n_cont ? cont_offsets[n_cont] : 0;
    m->max_chars = max_chars;
    m->cblob = (uint8_t*)malloc((size_t)total + 1);
    if (total) memcpy(m->cblob, cont_bytes, (size_t)total);
    m->cont = (piece_t*)malloc(sizeof(piece_t) * (size_t)(n_cont + 1));
    for (i = 0; i < n_cont; ++i) {
        piece_t* p = &m->cont[i];
        p->p = m->cblob + cont_offsets[i];
        p->len = cont_offsets[i + 1] - cont_offsets[i];
        p->id = cont_ids[i];
        p->score = 0.0;
        p->order = i;
        if (p->len > m->max_cont_len) m->max_cont_len = p->len;
    }
    qsort(m->cont, (size_t)n_cont, sizeof(piece_t), cmp_piece);
    m->n_cont = dedup_pieces(m->cont, n_cont);
}

void retok_ref_free(void* h) {
    model_t* m = (model_t*)h;
    if (!m) return;
    free(m->blob); free(m->pieces); free(m->merges); free(m->sblob); free(m->specials); free(m->cont); free(m->cblob); free(m);
}

/* ids of "<0xXX>" for every UTF-8 byte of the printable chars of raw[0..len); 0 if any is absent */
static int fallback_ids(const model_t* m, const uint8_t* raw, int len, int* out, int* n_out) {
    int i, k, n = 0;
    for (i = 0; i < len; ++i) {
        uint8_t u[2];
        int nu = char_utf8(raw[i], u);
        for (k = 0; k < nu; ++k) {
            int id = m->bf_ids[u[k]];
            if (id < 0) return 0;
            out[n++] = id;
        }
    }
    *n_out = n;
    return 1;
}

/* ---- BPE ---------------------------------------------------------------------------- */
typedef struct { int rank, pos, new_id; } qent_t;

static int bpe_tokenize(const model_t* m, const uint8_t* raw, int len, int* out /* cap 2*len+2 */) {
    int n = 0, i, nq = 0, cap;
    int *c, *prev, *next, *alive;
    qent_t* q;
    int unk_pending = 0;
    if (len == 0) return 0;
    if (m->ignore_merges) {
        const piece_t* p = find_piece(m->pieces, m->n_pieces, raw, len);
        if (p) { out[0] = p->id; return 1; }
    }
    cap = 2 * len + 2;
    c = (int*)malloc(sizeof(int) * (size_t)cap * 4);
    prev = c + cap; next = prev + cap; alive = next + cap;
    for (i = 0; i < len; ++i) {                       /* merge_word */
        int id = m->single_id[raw[i]];
        if (id >= 0) {
            if (unk_pending) { c[n++] = m->unk_id; unk_pending = 0; }
            c[n++] = id;
            continue;
        }
        if (m->byte_fallback) {
            int fb[2], nfb = 0;
            if (fallback_ids(m, raw + i, 1, fb, &nfb)) {
                int k;
                for (k = 0; k < nfb; ++k) c[n++] = fb[k];   /* a pending unk is NOT flushed first */
                continue;
            }
        }
        if (m->unk_id >= 0) {
            if (unk_pending && !m->fuse_unk) c[n++] = m->unk_id;
            unk_pending = 1;
        }
    }
    if (unk_pending) c[n++] = m->unk_id;
    for (i = 0; i < n; ++i) { prev[i] = i - 1; next[i] = (i + 1 < n) ? i + 1 : -1; alive[i] = 1; }
    q = (qent_t*)malloc(sizeof(qent_t) * (size_t)(3 * n + 4));
    for (i = 0; i + 1 < n; ++i) {                     /* Word::merge_all */
        const merge_t* e = find_merge(m, c[i], c[i + 1]);
        if (e) { q[nq].rank = e->rank; q[nq].pos = i; q[nq].new_id = e->new_id; ++nq; }
    }
    while (nq > 0) {
        int best = 0, pos, r;
        qent_t top;
        const merge_t* e;
        for (i = 1; i < nq; ++i)
            if (q[i].rank < q[best].rank || (q[i].rank == q[best].rank && q[i].pos < q[best].pos)) best = i;
        top = q[best];
        q[best] = q[--nq];
        pos = top.pos;
        if (!alive[pos] || next[pos] == -1) continue;
        r = next[pos];
        e = find_merge(m, c[pos], c[r]);
        if (!e || e->new_id != top.new_id) continue;   /* expired entry: compared by new id only */
        c[pos] = top.new_id;
        alive[r] = 0;
        next[pos] = next[r];
        if (next[r] != -1) prev[next[r]] = pos;
        if (prev[pos] >= 0) {
            e = find_merge(m, c[prev[pos]], c[pos]);
            if (e) { q[nq].rank = e->rank; q[nq].pos = prev[pos]; q[nq].new_id = e->new_id; ++nq; }
        }
        if (next[pos] != -1) {
            e = find_merge(m, c[pos], c[next[pos]]);
            if (e) { q[nq].rank = e->rank; q[nq].pos = pos; q[nq].new_id = e->new_id; ++nq; }
        }
    }
    {
        int w = 0;
        for (i = 0; i < n; ++i) if (alive[i]) out[w++] = c[i];
        n = w;
    }
    free(q); free(c);
    return n;
}

/* ---- Unigram ------------------------------------------------------------------------ */
static int unigram_tokenize(const model_t* m, const uint8_t* raw, int len, int* out /* cap 2*len+2 */) {
    double* best;
    int *bstart, *bid, *spans;
    int s, e, n_out = 0, ns = 0, i;
    const double unk_score = m->min_score - 10.0;      /* kUnkPenalty */
    if (len == 0) return 0;
    best = (double*)malloc(sizeof(double) * (size_t)(len + 1));
    bstart = (int*)malloc(sizeof(int) * (size_t)(len + 1) * 5);
    bid = bstart + (len + 1);
    spans = bid + (len + 1);                           /* triples (s, e, id) */
    for (i = 0; i <= len; ++i) { best[i] = 0.0; bstart[i] = -1; bid[i] = -1; }
    for (s = 0; s < len; ++s) {
        const double base = best[s];
        int has_single = 0;
        int emax = s + m->max_piece_len < len ? s + m->max_piece_len : len;
        for (e = s + 1; e <= emax; ++e) {
            const piece_t* p = find_piece(m->pieces, m->n_pieces, raw + s, e - s);
            double cand;
            if (!p) continue;
            cand = p->score + base;
            if (bstart[e] == -1 || cand > best[e]) { best[e] = cand; bstart[e] = s; bid[e] = p->id; }
            if (e == s + 1) has_single = 1;
        }
        if (!has_single) {
            double cand = unk_score + base;
            if (m->unk_id < 0) { free(best); free(bstart); return -1; }   /* "unk_id is missing" */
            e = s + 1;
            if (bstart[e] == -1 || cand > best[e]) { best[e] = cand; bstart[e] = s; bid[e] = -2; }
        }
    }
    for (e = len; e > 0; e = bstart[e]) { spans[3 * ns] = bstart[e]; spans[3 * ns + 1] = e; spans[3 * ns + 2] = bid[e]; ++ns; }
    for (i = ns - 1; i >= 0;) {
        if (spans[3 * i + 2] != -2) { out[n_out++] = spans[3 * i + 2]; --i; continue; }
        {   /* fuse the run of unknown pieces */
            int j = i, nfb = 0, ok = 0;
            int fs = spans[3 * i], fe;
            while (j - 1 >= 0 && spans[3 * (j - 1) + 2] == -2) --j;
            fe = spans[3 * j + 1];
            if (m->byte_fallback) ok = fallback_ids(m, raw + fs, fe - fs, out + n_out, &nfb);
            if (ok) n_out += nfb; else out[n_out++] = m->unk_id;
            i = j - 1;
        }
    }
    free(best); free(bstart);
    return n_out;
}

/* ---- WordPiece ----------------------------------------------------------------------
 * tokenizers `WordPiece::tokenize`: more than max_input_chars_per_word characters -> [unk]; otherwise, from each
 * start, the LONGEST substring that is in the vocabulary (with the continuing prefix when start > 0); no match at some
 * start -> the whole word is [unk].  A byte-level character is one raw byte, so characters = bytes here. */
static int wordpiece_tokenize(const model_t* m, const uint8_t* raw, int len, int* out /* cap 2*len+2 */) {
    int start = 0, n = 0;
    if (len == 0) return 0;
    if (len > m->max_chars) { if (m->unk_id < 0) return -1; out[0] = m->unk_id; return 1; }
    while (start < len) {
        int end = len;
        const piece_t* hit = NULL;
        while (start < end) {
            hit = start == 0 ? find_piece(m->pieces, m->n_pieces, raw, end) : find_piece(m->cont, m->n_cont, raw + start, end - start);
            if (hit) break;
            --end;
        }
        if (!hit) { if (m->unk_id < 0) return -1; out[0] = m->unk_id; return 1; }
        out[n++] = hit->id;
        start = end;
    }
    return n;
}

/* get_surface_form_matrix (zett/utils.py:651-689): `out` is pre-filled with pad_id by the caller */
int retok_ref_surface_forms(void* h, const uint8_t* raw, const int32_t* offsets, int64_t n_tokens, int maxlen,
                            int pad_id, int32_t* out, int64_t* n_truncated) {
    const model_t* m = (const model_t*)h;
    int64_t t, trunc = 0;
    (void)pad_id;
    for (t = 0; t < n_tokens; ++t) {
        const uint8_t* p = raw + offsets[t];
        int len = offsets[t + 1] - offsets[t];
        int32_t* row = out + t * maxlen;
        const piece_t* sp = len ? find_piece(m->specials, m->n_special, p, len) : NULL;
        int* ids;
        int n, i;
        if (sp) { row[0] = sp->id; continue; }          /* :671-673 */
        ids = (int*)malloc(sizeof(int) * (size_t)(2 * len + 4));
        n = m->kind == 0 ? bpe_tokenize(m, p, len, ids) : m->kind == 1 ? unigram_tokenize(m, p, len, ids) : wordpiece_tokenize(m, p, len, ids);
        if (n < 0) { free(ids); return -1; }
        if (n > maxlen) { n = maxlen; ++trunc; }        /* :683-685 */
        for (i = 0; i < n; ++i) row[i] = ids[i];
        free(ids);
    }
    *n_truncated = trunc;
    return 0;
}
