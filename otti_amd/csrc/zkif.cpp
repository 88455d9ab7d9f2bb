// zkInterface ingest for the spzk boundary: a hand-written, bounds-checked FlatBuffers reader (and a writer used to emit
// synthetic instances as real .zkif triples).  Replaces the `zkinterface` crate + R1CS builder inside spartan-zkinterface
// [RECALL; /root/reference/spartan-zkinterface is an empty submodule]; the three-file split follows
// /root/reference/run.py:47-49 (X.zkif = header + constraints, X.inp.zkif = header with instance values, X.wit.zkif = witness).
//
// Schema (zkinterface 1.x): Root{message: union{CircuitHeader=1, ConstraintSystem=2, Witness=3, Command=4}};
// CircuitHeader{instance_variables: Variables, free_variable_id: u64, field_maximum: [u8], configuration};
// ConstraintSystem{constraints: [BilinearConstraint{linear_combination_a, _b, _c: Variables}]}; Witness{assigned_variables: Variables};
// Variables{variable_ids: [u64], values: [u8], info}.  Files are concatenations of size-prefixed buffers with identifier "zkif".
#include "spartan.h"
#include <stdio.h>
#include <stdlib.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <map>
#include <algorithm>
#include <thread>
#include <exception>
#include <system_error>
#include <new>
#include <chrono>

namespace otti {

namespace {
struct Buf {
    const uint8_t *p; size_t n;
    void chk(size_t off, size_t len) const { if (off > n || len > n - off) throw Error(OTTI_ERR_IO, "zkif: offset out of bounds"); }
    uint8_t u8(size_t o) const { chk(o, 1); return p[o]; }
    // FlatBuffers are little-endian, and so is every host this library is built for (x86-64): plain unaligned loads after ONE bounds test
    uint16_t u16(size_t o) const { chk(o, 2); uint16_t v; memcpy(&v, p + o, 2); return v; }
    uint32_t u32(size_t o) const { chk(o, 4); uint32_t v; memcpy(&v, p + o, 4); return v; }
    int32_t i32(size_t o) const { return (int32_t)u32(o); }
    uint64_t u64(size_t o) const { chk(o, 8); uint64_t v; memcpy(&v, p + o, 8); return v; }
};
static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "the zkInterface reader loads little-endian fields directly");
struct Table {
    const Buf *b; size_t pos = 0; bool present = false;
    // absolute position of field `id`, or 0 when absent
    size_t field(int id) const {
        if (!present) return 0;
        int64_t vt = (int64_t)pos - b->i32(pos);
        if (vt < 0) throw Error(OTTI_ERR_IO, "zkif: bad vtable offset");
        uint16_t vsize = b->u16((size_t)vt);
        size_t slot = 4 + 2 * (size_t)id;
        if (slot + 2 > vsize) return 0;
        uint16_t off = b->u16((size_t)vt + slot);
        return off ? pos + off : 0;
    }
    Table sub(int id) const { Table t; t.b = b; size_t f = field(id); if (f) { t.pos = f + b->u32(f); t.present = true; b->chk(t.pos, 4); } return t; }
    // vector field: returns element start and count
    bool vec(int id, size_t &start, size_t &count) const {
        size_t f = field(id); if (!f) { start = count = 0; return false; }
        size_t v = f + b->u32(f); count = b->u32(v); start = v + 4; return true;
    }
    uint64_t u64f(int id, uint64_t dflt) const { size_t f = field(id); return f ? b->u64(f) : dflt; }
    uint8_t u8f(int id, uint8_t dflt) const { size_t f = field(id); return f ? b->u8(f) : dflt; }
};
struct VarList { std::vector<uint64_t> ids; std::vector<std::array<uint8_t, 32>> vals; bool has_vals = false; };

VarList read_variables(const Table &t) {
    VarList v; if (!t.present) return v;
    size_t s, n; t.vec(0, s, n);
    t.b->chk(s, n * 8);
    v.ids.resize(n); for (size_t i = 0; i < n; i++) v.ids[i] = t.b->u64(s + 8 * i);
    size_t vs, vn;
    if (t.vec(1, vs, vn) && vn) {
        if (n == 0 || vn % n) throw Error(OTTI_ERR_IO, "zkif: values length not a multiple of the variable count");
        size_t w = vn / n; t.b->chk(vs, vn);
        v.vals.resize(n); v.has_vals = true;
        if (w == 32) memcpy(v.vals.data(), t.b->p + vs, vn);                  // the common case: one block copy
        else for (size_t i = 0; i < n; i++) {
            v.vals[i].fill(0);
            for (size_t k = 0; k < w; k++) {
                uint8_t byte = t.b->p[vs + i * w + k];
                if (k < 32) v.vals[i][k] = byte; else if (byte) throw Error(OTTI_ERR_INVALID_SCALAR, "zkif: element wider than 32 bytes");
            }
        }
    }
    return v;
}

// A .zkif file as a read-only memory map: multi-GB constraint files (2^24 constraints) are paged in by the kernel as the two passes
// walk them, never copied into the process heap.
struct FileView {
    const uint8_t *p = nullptr; size_t n = 0; int fd = -1;
    explicit FileView(const char *path) {
        fd = open(path, O_RDONLY);
        if (fd < 0) throw Error(OTTI_ERR_IO, std::string("cannot open ") + path);
        struct stat st;
        if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); throw Error(OTTI_ERR_IO, std::string("not a regular file: ") + path); }
        n = (size_t)st.st_size;
        if (n) {
            // MAP_POPULATE: the parser's threads walk the file in parallel; one batched population of the mapping is several times
            // cheaper than a minor fault per 4 KiB page taken by sixteen threads contending for the address-space lock.  It blocks
            // until the whole file is resident, though: above kPopulateMax the mapping is only advised (read-ahead) and paged in as
            // the passes walk it.
            constexpr size_t kPopulateMax = (size_t)2 << 30;
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE | (n <= kPopulateMax ? MAP_POPULATE : 0), fd, 0);
            if (m == MAP_FAILED) { close(fd); throw Error(OTTI_ERR_IO, std::string("cannot map ") + path); }
            if (n > kPopulateMax) (void)madvise(m, n, MADV_WILLNEED);
            p = (const uint8_t *)m;
        }
    }
    FileView(const FileView &) = delete; FileView &operator=(const FileView &) = delete;
    ~FileView() { if (p) munmap((void *)p, n); if (fd >= 0) close(fd); }
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
};

struct Messages {
    bool have_header = false; VarList instance; uint64_t free_variable_id = 0; std::vector<uint8_t> field_maximum;
    VarList witness; bool have_witness = false;
};
// walks the size-prefixed messages of one file; fn(type, message table, buffer)
template <class F> void for_each_message(const FileView &file, F &&fn) {
    size_t off = 0;
    while (off + 4 <= file.size()) {
        const uint8_t *q = file.data() + off; uint32_t sz = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
        if (sz < 8 || sz > file.size() - off - 4) throw Error(OTTI_ERR_IO, "zkif: bad message size prefix");
        Buf b{file.data() + off + 4, sz};
        if (memcmp(b.p + 4, "zkif", 4) != 0) throw Error(OTTI_ERR_IO, "zkif: missing file identifier");
        Table root; root.b = &b; root.pos = b.u32(0); root.present = true; b.chk(root.pos, 4);
        fn(root.u8f(0, 0), root.sub(1), b);
        off += 4 + sz;
    }
    if (off != file.size()) throw Error(OTTI_ERR_IO, "zkif: trailing bytes");
}
// pass 1: headers and witness only (ConstraintSystem messages are skipped by their size prefix)
void parse_headers(const FileView &file, Messages &m) {
    for_each_message(file, [&](uint8_t type, const Table &msg, const Buf &b) {
        if (type == 1) {                       // CircuitHeader
            VarList iv = read_variables(msg.sub(0));
            if (!m.have_header || iv.has_vals) m.instance = iv;           // the .inp.zkif header carries the values
            m.free_variable_id = std::max(m.free_variable_id, msg.u64f(1, 0));
            size_t s, n; if (msg.vec(2, s, n)) { b.chk(s, n); m.field_maximum.assign(b.p + s, b.p + s + n); }
            m.have_header = true;
        } else if (type == 3) {                // Witness
            VarList w = read_variables(msg.sub(0));
            m.witness.ids.insert(m.witness.ids.end(), w.ids.begin(), w.ids.end());
            m.witness.vals.insert(m.witness.vals.end(), w.vals.begin(), w.vals.end());
            m.have_witness = true;
        }                                      // ConstraintSystem / Command / unknown: not here
    });
}

// ---- forward-laid-out FlatBuffers writer (every uoffset points forward; vtables sit right before their tables)
struct Fbw {
    std::vector<uint8_t> b;
    void pad_to(size_t align, size_t bias = 0) { while ((b.size() + bias) % align) b.push_back(0); }
    void u16(uint16_t x) { b.push_back((uint8_t)x); b.push_back((uint8_t)(x >> 8)); }
    void u32(uint32_t x) { for (int i = 0; i < 4; i++) b.push_back((uint8_t)(x >> (8 * i))); }
    void u64(uint64_t x) { for (int i = 0; i < 8; i++) b.push_back((uint8_t)(x >> (8 * i))); }
    void patch(size_t at, size_t target) { uint32_t v = (uint32_t)(target - at); for (int i = 0; i < 4; i++) b[at + i] = (uint8_t)(v >> (8 * i)); }
    // table with fields given as (size in {1,4,8}, inline value); size 4 = offset placeholder.  Returns table pos and placeholder positions.
    struct F { int size; uint64_t v; };
    size_t table(const std::vector<F> &fs, std::vector<size_t> &slots) {
        std::vector<uint16_t> off(fs.size()); size_t cur = 4;
        for (size_t i = 0; i < fs.size(); i++) { if (fs[i].size == 0) { off[i] = 0; continue; } cur = (cur + fs[i].size - 1) / fs[i].size * fs[i].size; off[i] = (uint16_t)cur; cur += fs[i].size; }
        size_t vsize = 4 + 2 * fs.size();
        pad_to(8, vsize);                      // table start 8-aligned
        size_t vt = b.size();
        u16((uint16_t)vsize); u16((uint16_t)cur); for (auto o : off) u16(o);
        size_t tp = b.size();
        u32((uint32_t)(tp - vt));
        slots.assign(fs.size(), 0);
        size_t written = 4;
        for (size_t i = 0; i < fs.size(); i++) {
            if (!fs[i].size) continue;
            while (written < off[i]) { b.push_back(0); written++; }
            if (fs[i].size == 1) b.push_back((uint8_t)fs[i].v); else if (fs[i].size == 4) { slots[i] = b.size(); u32(0); } else u64(fs[i].v);
            written += fs[i].size;
        }
        return tp;
    }
    size_t vec_u64(const std::vector<uint64_t> &v) { pad_to(8, 4); size_t p = b.size(); u32((uint32_t)v.size()); for (auto x : v) u64(x); return p; }
    size_t vec_u8(const uint8_t *d, size_t n) { pad_to(4); size_t p = b.size(); u32((uint32_t)n); b.insert(b.end(), d, d + n); return p; }
    size_t variables(const std::vector<uint64_t> &ids, const std::vector<uint8_t> &values) {
        std::vector<size_t> sl; size_t tp = table({{4, 0}, {4, 0}}, sl);
        patch(sl[0], vec_u64(ids)); patch(sl[1], vec_u8(values.data(), values.size()));
        return tp;
    }
};
void begin_message(Fbw &w) { w.b.clear(); w.u32(0); w.b.insert(w.b.end(), {'z', 'k', 'i', 'f'}); }
void finish_message(Fbw &w, size_t root_pos, FILE *f) {
    w.patch(0, root_pos);                      // root uoffset is relative to position 0
    w.pad_to(8);
    uint32_t sz = (uint32_t)w.b.size(); uint8_t pre[4] = {(uint8_t)sz, (uint8_t)(sz >> 8), (uint8_t)(sz >> 16), (uint8_t)(sz >> 24)};
    if (fwrite(pre, 1, 4, f) != 4 || fwrite(w.b.data(), 1, w.b.size(), f) != w.b.size()) throw Error(OTTI_ERR_IO, "zkif: write failed");
}
size_t root_with(Fbw &w, uint8_t type, size_t &msg_slot) {
    std::vector<size_t> sl; size_t tp = w.table({{1, type}, {4, 0}}, sl); msg_slot = sl[1]; return tp;
}
void write_header(FILE *f, const std::vector<uint64_t> &inst_ids, const std::vector<uint8_t> &inst_vals, uint64_t free_id) {
    static const uint8_t lm1[32] = {0xec, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7, 0xa2, 0xde, 0xf9, 0xde, 0x14,
                                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x10};
    Fbw w; begin_message(w);
    size_t slot; size_t root = root_with(w, 1, slot);
    std::vector<size_t> sl; size_t hdr = w.table({{4, 0}, {8, free_id}, {4, 0}}, sl);
    w.patch(slot, hdr);
    w.patch(sl[0], w.variables(inst_ids, inst_vals));
    w.patch(sl[2], w.vec_u8(lm1, 32));
    finish_message(w, root, f);
}
}  // namespace

// column of a zkInterface variable id in Spartan's z = [vars | 1 | inputs]; ids are dense in [0, free_variable_id), so a flat table
struct IdMap {
    std::vector<uint32_t> col_of;             // 0xffffffff = undeclared
    size_t num_vars = 0, num_inputs = 0;
    size_t col(uint64_t id) const {
        if (id >= col_of.size() || col_of[id] == 0xffffffffu) throw Error(OTTI_ERR_INVALID_INDEX, "zkif: constraint references an undeclared variable id");
        return col_of[id];
    }
    // witness position of an id (its column while it is below num_vars), or SIZE_MAX
    size_t wit_pos(uint64_t id) const { if (id >= col_of.size()) return SIZE_MAX; uint32_t c = col_of[id]; return c < num_vars ? c : SIZE_MAX; }
    void build(const std::vector<uint64_t> &instance_ids, uint64_t free_id) {
        if (free_id > ((uint64_t)1 << 31)) throw Error(OTTI_ERR_BAD_ARG, "zkif: more than 2^31 variable ids");
        for (auto id : instance_ids) if (id == 0 || id >= free_id) throw Error(OTTI_ERR_INVALID_INDEX, "zkif: instance variable id out of range");
        num_inputs = instance_ids.size();
        std::vector<uint8_t> is_inst((size_t)free_id, 0);
        for (auto id : instance_ids) { if (is_inst[id]) throw Error(OTTI_ERR_IO, "zkif: duplicate instance variable id"); is_inst[id] = 1; }
        num_vars = free_id > 0 ? (size_t)free_id - 1 - num_inputs : 0;
        col_of.assign((size_t)std::max<uint64_t>(free_id, 1), 0xffffffffu);
        col_of[0] = (uint32_t)num_vars;                                     // id 0 is the constant one
        for (size_t i = 0; i < instance_ids.size(); i++) col_of[instance_ids[i]] = (uint32_t)(num_vars + 1 + i);
        uint32_t k = 0;
        for (uint64_t id = 1; id < free_id; id++) if (!is_inst[id]) col_of[id] = k++;
    }
};

}  // namespace otti

using namespace otti;

static otti_entry *copy_entries(const std::vector<otti_entry> &v) {
    otti_entry *p = (otti_entry *)malloc(std::max<size_t>(1, v.size()) * sizeof(otti_entry));
    if (!v.empty()) memcpy(p, v.data(), v.size() * sizeof(otti_entry));
    return p;
}
static uint8_t *copy_bytes(const std::vector<uint8_t> &v) {
    uint8_t *p = (uint8_t *)malloc(std::max<size_t>(1, v.size()));
    if (!v.empty()) memcpy(p, v.data(), v.size());
    return p;
}

// run f(0 .. nt) on nt threads (the caller's included).  A thread that cannot be started (EAGAIN under a pids cgroup or
// RLIMIT_NPROC) must not unwind through joinable std::threads (std::terminate): the shares that found no thread run here instead.
template <class F> static void fan_out(unsigned nt, F &&f) {
    std::vector<std::thread> th; th.reserve(nt);
    unsigned started = 1;
    for (; started < nt; started++) {
        try { th.emplace_back(f, started); } catch (const std::system_error &) { break; }
    }
    f(0u);
    for (unsigned t = started; t < nt; t++) f(t);
    for (auto &x : th) x.join();
}

otti_r1cs *otti_r1cs_from(size_t nc, size_t nv, size_t ni, const std::vector<otti_entry> &A, const std::vector<otti_entry> &B,
                          const std::vector<otti_entry> &C, const std::vector<uint8_t> &vars, const std::vector<uint8_t> &inputs) {
    otti_r1cs *r = (otti_r1cs *)calloc(1, sizeof *r);
    if (!r) throw std::bad_alloc();
    r->num_cons = nc; r->num_vars = nv; r->num_inputs = ni;
    r->A = copy_entries(A); r->nA = A.size(); r->B = copy_entries(B); r->nB = B.size(); r->C = copy_entries(C); r->nC = C.size();
    r->vars32 = copy_bytes(vars); r->nvars = vars.size() / 32; r->inputs32 = copy_bytes(inputs); r->ninputs = inputs.size() / 32;
    return r;
}

// one linear combination (a Variables table) straight into matrix entries of `row`; no intermediate per-constraint objects, so a
// 2^24-constraint file costs its own size plus the entry arrays
static void append_lc(std::vector<otti_entry> &out, const Table &vars, uint64_t row, const IdMap &map) {
    if (!vars.present) return;
    size_t is, in; vars.vec(0, is, in);
    if (!in) return;
    vars.b->chk(is, in * 8);
    size_t vs, vn;
    if (!vars.vec(1, vs, vn) || vn == 0 || vn % in) throw Error(OTTI_ERR_IO, "zkif: linear combination without (well-formed) coefficients");
    const size_t w = vn / in; vars.b->chk(vs, vn);
    for (size_t i = 0; i < in; i++) {
        otti_entry e; e.row = row; e.col = map.col(vars.b->u64(is + 8 * i));
        const uint8_t *src = vars.b->p + vs + i * w;
        if (w == 32) memcpy(e.val, src, 32);
        else {
            memset(e.val, 0, 32); memcpy(e.val, src, std::min<size_t>(w, 32));
            for (size_t k = 32; k < w; k++) if (src[k]) throw Error(OTTI_ERR_INVALID_SCALAR, "zkif: coefficient wider than 32 bytes");
        }
        out.push_back(e);
    }
}

static double zk_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
otti_r1cs *zkif_load_impl(const char *circuit_path, const char *inputs_path, const char *witness_path) {
    const bool trace = getenv("OTTI_TRACE") != nullptr; double tt = zk_now();
    auto lap = [&](const char *what) { if (trace) { double n = zk_now(); fprintf(stderr, "[otti] zkif_load %-28s %.2f ms\n", what, n - tt); tt = n; } };
    Messages m;
    FileView circuit(circuit_path);
    parse_headers(circuit, m);
    size_t file_bytes = circuit.size();
    if (inputs_path) { FileView f(inputs_path); parse_headers(f, m); file_bytes += f.size(); }
    if (witness_path) { FileView f(witness_path); parse_headers(f, m); file_bytes += f.size(); }
    if (!m.have_header) throw Error(OTTI_ERR_IO, "zkif: no CircuitHeader message");
    if (!m.field_maximum.empty()) {
        static const uint8_t lm1[32] = {0xec, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7, 0xa2, 0xde, 0xf9, 0xde, 0x14,
                                        0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x10};
        std::vector<uint8_t> fm = m.field_maximum; while (!fm.empty() && fm.back() == 0) fm.pop_back();
        std::vector<uint8_t> want(lm1, lm1 + 32); while (!want.empty() && want.back() == 0) want.pop_back();
        if (fm != want) throw Error(OTTI_ERR_IO, "zkif: field_maximum is not l-1 for the curve25519 scalar field");
    }
    // witness variables: every id in [1, free_variable_id) that is not an instance variable, in increasing order
    uint64_t free_id = std::max<uint64_t>(m.free_variable_id, 1);
    for (auto id : m.witness.ids) free_id = std::max(free_id, id + 1);
    for (auto id : m.instance.ids) free_id = std::max(free_id, id + 1);
    // Untrusted header: every variable id a file declares occupies at least eight bytes somewhere in the files (its assignment, or its
    // uses in constraints), so a free_variable_id beyond that is either garbage or an attempt to make the loader allocate gigabytes.
    if (free_id > file_bytes / 8 + 2) throw Error(OTTI_ERR_IO, "zkif: free_variable_id is larger than the files can account for");
    lap("headers + witness message");
    IdMap map; map.build(m.instance.ids, free_id);
    lap("id map");
    if (m.have_witness && m.witness.ids.size() != map.num_vars) throw Error(OTTI_ERR_IO, "zkif: witness does not assign every variable exactly once");
    // pass 2: constraints of the circuit file.  A quick walk over the messages finds every ConstraintSystem's offset vector; the rows are
    // then cut into contiguous blocks, one per host thread, each parsed into its own entry lists (a FlatBuffers vector of tables is
    // random access), and the lists are concatenated in row order — the same entries in the same order as a sequential walk.
    struct Chunk { Buf b; size_t s, n; uint64_t first_row; };
    std::vector<Chunk> chunks; uint64_t row = 0;
    for_each_message(circuit, [&](uint8_t type, const Table &msg, const Buf &b) {
        if (type != 2) return;
        size_t s, n; msg.vec(0, s, n); b.chk(s, n * 4);
        chunks.push_back({b, s, n, row}); row += n;
    });
    lap("message walk");
    unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (const char *e = getenv("OTTI_PARSE_THREADS")) { int v = atoi(e); if (v >= 1 && v <= 64) nt = (unsigned)v; }
    if (row < 4096) nt = 1;
    std::vector<std::vector<otti_entry>> part[3]; for (auto &p : part) p.resize(nt);
    std::vector<std::exception_ptr> err(nt);
    auto work = [&](unsigned t) {
        try {
            const uint64_t r0 = row * t / nt, r1 = row * (t + 1) / nt;
            for (auto &v : part) v[t].reserve((size_t)((r1 - r0) * 5 / 4 + 16));
            for (const Chunk &ch : chunks) {
                if (ch.first_row + ch.n <= r0 || ch.first_row >= r1) continue;
                const size_t i0 = r0 > ch.first_row ? (size_t)(r0 - ch.first_row) : 0, i1 = (size_t)std::min<uint64_t>(ch.n, r1 - ch.first_row);
                for (size_t i = i0; i < i1; i++) {
                    Table bc; bc.b = &ch.b; bc.pos = ch.s + 4 * i + ch.b.u32(ch.s + 4 * i); bc.present = true; ch.b.chk(bc.pos, 4);
                    for (int k = 0; k < 3; k++) append_lc(part[k][t], bc.sub(k), ch.first_row + i, map);
                }
            }
        } catch (...) { err[t] = std::current_exception(); }
    };
    fan_out(nt, work);
    for (auto &e : err) if (e) std::rethrow_exception(e);
    lap("constraints (threads)");
    otti_r1cs *out = (otti_r1cs *)calloc(1, sizeof *out);
    if (!out) throw std::bad_alloc();
    struct Guard { otti_r1cs *r; ~Guard() { if (r) otti_r1cs_free(r); } } guard{out};
    otti_entry **dst[3] = {&out->A, &out->B, &out->C}; size_t *cnt[3] = {&out->nA, &out->nB, &out->nC};
    for (int k = 0; k < 3; k++) {                                         // the threads' parts go straight into the arrays the caller will own
        size_t total = 0; for (auto &v : part[k]) total += v.size();
        otti_entry *arr = (otti_entry *)malloc(std::max<size_t>(1, total) * sizeof(otti_entry));
        if (!arr) throw std::bad_alloc();
        *dst[k] = arr; *cnt[k] = total;
        std::vector<size_t> at(nt); size_t o = 0; for (unsigned t = 0; t < nt; t++) { at[t] = o; o += part[k][t].size(); }
        auto cp = [&, k, arr](unsigned t) { if (!part[k][t].empty()) memcpy(arr + at[t], part[k][t].data(), part[k][t].size() * sizeof(otti_entry)); std::vector<otti_entry>().swap(part[k][t]); };
        fan_out(nt, cp);
    }
    lap("concatenate");
    out->num_cons = row; out->num_vars = map.num_vars; out->num_inputs = map.num_inputs;
    out->nvars = m.have_witness ? map.num_vars : 0; out->ninputs = map.num_inputs;      // no witness file (verifier): no assignment
    out->vars32 = (uint8_t *)calloc(std::max<size_t>(1, 32 * out->nvars), 1); out->inputs32 = (uint8_t *)calloc(std::max<size_t>(1, 32 * out->ninputs), 1);
    if (!out->vars32 || !out->inputs32) throw std::bad_alloc();
    if (m.instance.has_vals) for (size_t i = 0; i < map.num_inputs; i++) memcpy(out->inputs32 + 32 * i, m.instance.vals[i].data(), 32);
    else if (map.num_inputs && inputs_path) throw Error(OTTI_ERR_IO, "zkif: inputs file carries no instance values");
    if (m.have_witness) {
        if (m.witness.ids.size() != m.witness.vals.size()) throw Error(OTTI_ERR_IO, "zkif: witness without values");
        std::vector<uint8_t> seen(map.num_vars, 0);
        for (size_t i = 0; i < m.witness.ids.size(); i++) {
            size_t pos = m.witness.ids[i] == 0 ? SIZE_MAX : map.wit_pos(m.witness.ids[i]);
            if (pos == SIZE_MAX) throw Error(OTTI_ERR_INVALID_INDEX, "zkif: witness assigns an instance or unknown variable");
            memcpy(out->vars32 + 32 * pos, m.witness.vals[i].data(), 32); seen[pos] = 1;
        }
        for (uint8_t sn : seen) if (!sn) throw Error(OTTI_ERR_IO, "zkif: witness does not assign every variable");
    }
    lap("assignment");
    guard.r = nullptr;
    return out;
}

void zkif_write_impl(const otti_r1cs *r, const char *circuit_path, const char *inputs_path, const char *witness_path) {
    // ids: 0 = one, 1..ni = instance variables, ni+1.. = witness variables
    const size_t ni = r->num_inputs, nv = r->num_vars;
    auto id_of_col = [&](uint64_t col) -> uint64_t { return col < nv ? ni + 1 + col : col == nv ? 0 : col - nv; };
    std::vector<uint64_t> inst_ids(ni); for (size_t i = 0; i < ni; i++) inst_ids[i] = i + 1;
    uint64_t free_id = ni + nv + 1;
    FILE *f = fopen(circuit_path, "wb"); if (!f) throw Error(OTTI_ERR_IO, "zkif: cannot create circuit file");
    try {
        write_header(f, inst_ids, {}, free_id);
        // constraints, grouped by row, in chunks of 2^16 rows per ConstraintSystem message
        const otti_entry *src[3] = {r->A, r->B, r->C}; size_t cnt[3] = {r->nA, r->nB, r->nC};
        std::vector<std::vector<std::pair<uint64_t, const uint8_t *>>> rows[3];
        for (int t = 0; t < 3; t++) { rows[t].resize(r->num_cons); for (size_t i = 0; i < cnt[t]; i++) rows[t][src[t][i].row].push_back({id_of_col(src[t][i].col), src[t][i].val}); }
        const size_t chunk = 1 << 16;
        for (size_t r0 = 0; r0 < r->num_cons || r0 == 0; r0 += chunk) {
            size_t r1 = std::min<size_t>(r->num_cons, r0 + chunk);
            Fbw w; begin_message(w);
            size_t slot; size_t root = root_with(w, 2, slot);
            std::vector<size_t> sl; size_t cs = w.table({{4, 0}}, sl); w.patch(slot, cs);
            w.pad_to(4); size_t vpos = w.b.size(); w.patch(sl[0], vpos);
            w.u32((uint32_t)(r1 - r0)); size_t elems = w.b.size();
            for (size_t i = r0; i < r1; i++) w.u32(0);
            for (size_t i = r0; i < r1; i++) {
                std::vector<size_t> s3; size_t bc = w.table({{4, 0}, {4, 0}, {4, 0}}, s3);
                w.patch(elems + 4 * (i - r0), bc);
                for (int t = 0; t < 3; t++) {
                    std::vector<uint64_t> ids; std::vector<uint8_t> vals;
                    for (auto &e : rows[t][i]) { ids.push_back(e.first); vals.insert(vals.end(), e.second, e.second + 32); }
                    w.patch(s3[t], w.variables(ids, vals));
                }
            }
            finish_message(w, root, f);
            if (r->num_cons == 0) break;
        }
        fclose(f); f = nullptr;
        f = fopen(inputs_path, "wb"); if (!f) throw Error(OTTI_ERR_IO, "zkif: cannot create inputs file");
        write_header(f, inst_ids, std::vector<uint8_t>(r->inputs32, r->inputs32 + 32 * ni), free_id);
        fclose(f); f = nullptr;
        f = fopen(witness_path, "wb"); if (!f) throw Error(OTTI_ERR_IO, "zkif: cannot create witness file");
        {
            Fbw w; begin_message(w);
            size_t slot; size_t root = root_with(w, 3, slot);
            std::vector<size_t> sl; size_t wt = w.table({{4, 0}}, sl); w.patch(slot, wt);
            std::vector<uint64_t> ids(nv); for (size_t i = 0; i < nv; i++) ids[i] = ni + 1 + i;
            w.patch(sl[0], w.variables(ids, std::vector<uint8_t>(r->vars32, r->vars32 + 32 * nv)));
            finish_message(w, root, f);
        }
        fclose(f); f = nullptr;
    } catch (...) { if (f) fclose(f); throw; }
}
