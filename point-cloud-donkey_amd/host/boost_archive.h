// boost_archive.h — reader / writer for the byte stream of boost::archive::binary_oarchive, without Boost.
// The reference persists its trained data (.ismd) through `oa << x` of primitives, std::string and std::vector<float|unsigned>
// only (utils/json_object.cpp:84-87; implicit_shape_model.cpp:1144-1237, codebook/codebook.cpp:739-950, codeword.cpp:71-100,
// codeword_distribution.cpp:349-465, voting/voting.cpp:559-734): no class-typed objects, hence no class-id / tracking /
// version records in the stream. Layout (x86-64, little endian; EXTERNAL: Boost.Serialization, restated from knowledge of
// Boost 1.71, the version of the reference's target system -- no .ismd file ships with the reference, so the FORMAT IS
// UNPINNED until a file written by the real reference has been read):
//   header   : u64 22, "serialization::archive", u16 library version (17 for Boost 1.71),
//              u8 sizeof(int) = 4, u8 sizeof(long) = 8, u8 sizeof(float) = 4, u8 sizeof(double) = 8, i32 1 (endianness probe)
//   int / unsigned / float : 4 raw bytes;  bool : 1;  double : 8
//   std::string            : u64 length + bytes
//   std::vector<float|unsigned> (object_serializable, array-optimised since library version 6): u64 count + count * 4 raw bytes
#pragma once
#include <cstdint>
#include <cstring>
#include <istream>
#include <ostream>
#include <string>
#include <vector>

namespace ism3d {

class BoostBinaryOArchive {
public:
    explicit BoostBinaryOArchive(std::ostream& os, uint16_t library_version = 17) : m_os(os) {
        const std::string sig = "serialization::archive";
        raw<uint64_t>(sig.size()); m_os.write(sig.data(), (std::streamsize)sig.size());
        raw<uint16_t>(library_version);
        raw<uint8_t>(4); raw<uint8_t>(8); raw<uint8_t>(4); raw<uint8_t>(8);
        raw<int32_t>(1);
    }
    BoostBinaryOArchive& operator<<(int v) { raw<int32_t>(v); return *this; }
    BoostBinaryOArchive& operator<<(unsigned v) { raw<uint32_t>(v); return *this; }
    BoostBinaryOArchive& operator<<(float v) { raw<float>(v); return *this; }
    BoostBinaryOArchive& operator<<(const std::string& s) { raw<uint64_t>(s.size()); m_os.write(s.data(), (std::streamsize)s.size()); return *this; }
    template <typename T> BoostBinaryOArchive& operator<<(const std::vector<T>& v) {
        static_assert(sizeof(T) == 4, "only std::vector<float|unsigned|int> are streamed by the reference");
        raw<uint64_t>(v.size());
        if (!v.empty()) m_os.write((const char*)v.data(), (std::streamsize)(v.size() * 4));
        return *this;
    }
    bool good() const { return (bool)m_os; }
private:
    template <typename T> void raw(T v) { m_os.write((const char*)&v, sizeof(T)); }
    std::ostream& m_os;
};

class BoostBinaryIArchive {
public:
    // sets ok() = false (with a reason) instead of throwing: a wrong or cut file is an error return of readObject
    explicit BoostBinaryIArchive(std::istream& is) : m_is(is) {
        m_is.seekg(0, std::ios::end); m_size = (uint64_t)m_is.tellg(); m_is.seekg(0, std::ios::beg);
        uint64_t n = 0; raw(n);
        char sig[22] = {0};
        if (!m_ok || n != 22) { fail("not a Boost binary archive (signature length)"); return; }
        m_is.read(sig, 22);
        if (!m_is || std::memcmp(sig, "serialization::archive", 22) != 0) { fail("not a Boost binary archive (signature)"); return; }
        raw(m_version);
        uint8_t si = 0, sl = 0, sf = 0, sd = 0; int32_t one = 0;
        raw(si); raw(sl); raw(sf); raw(sd); raw(one);
        if (!m_ok) return;
        if (m_version < 6) { fail("Boost archive library version < 6 (vector framing differs) is not supported"); return; }
        if (si != 4 || sl != 8 || sf != 4 || sd != 8 || one != 1) fail("archive written on an incompatible platform (type sizes / endianness)");
    }
    BoostBinaryIArchive& operator>>(int& v) { int32_t t = 0; raw(t); v = t; return *this; }
    BoostBinaryIArchive& operator>>(unsigned& v) { uint32_t t = 0; raw(t); v = t; return *this; }
    BoostBinaryIArchive& operator>>(float& v) { raw(v); return *this; }
    BoostBinaryIArchive& operator>>(std::string& s) {
        uint64_t n = 0; raw(n);
        if (!m_ok || n > remaining()) { fail("string length exceeds the file"); s.clear(); return *this; }
        s.resize((size_t)n);
        if (n) m_is.read(&s[0], (std::streamsize)n);
        if (!m_is) fail("unexpected end of file");
        return *this;
    }
    template <typename T> BoostBinaryIArchive& operator>>(std::vector<T>& v) {
        static_assert(sizeof(T) == 4, "only 4-byte element vectors");
        uint64_t n = 0; raw(n);
        if (!m_ok || n > remaining() / 4) { fail("vector length exceeds the file"); v.clear(); return *this; }   // never resize to an unchecked length
        v.resize((size_t)n);
        if (n) m_is.read((char*)v.data(), (std::streamsize)(n * 4));
        if (!m_is) fail("unexpected end of file");
        return *this;
    }
    // a count that is about to drive a loop of at least `min_bytes_each` bytes per item
    bool plausible(uint64_t count, uint64_t min_bytes_each) { if (!m_ok) return false; if (count > remaining() / (min_bytes_each ? min_bytes_each : 1)) { fail("element count exceeds the file"); return false; } return true; }
    bool ok() const { return m_ok; }
    const std::string& error() const { return m_err; }
    uint16_t libraryVersion() const { return m_version; }
    void fail(const std::string& why) { if (m_ok) { m_ok = false; m_err = why; } }
private:
    uint64_t remaining() { const std::streampos p = m_is.tellg(); return p < 0 ? 0 : m_size - (uint64_t)p; }
    template <typename T> void raw(T& v) { if (!m_ok) return; m_is.read((char*)&v, sizeof(T)); if (!m_is) fail("unexpected end of file"); }
    std::istream& m_is; uint64_t m_size = 0; bool m_ok = true; std::string m_err; uint16_t m_version = 0;
};

}  // namespace ism3d
