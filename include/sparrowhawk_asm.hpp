// sparrowhawk_asm.hpp — header-only C++ mirror of the reference's AssemblyHelper
// (/root/reference/www/src/workers/Assembler.ts:15-39) over the C ABI in shk.h.  Same five
// members, same argument order; throws std::runtime_error where the Rust crate panics
// (Assembler.ts:93-106 catches that as a JS exception).
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include "shk.h"

namespace sparrowhawk {

class AssemblyHelper {
public:
    // AssemblyHelper::new — a static factory in the reference, not a constructor (Assembler.ts:94)
    static AssemblyHelper new_(uint32_t k, bool verbose, uint32_t min_count, uint32_t min_qual,
                               uint64_t chunk_size, bool do_bloom, bool do_fit,
                               bool no_bubble_collapse, bool no_dead_end_removal) {
        shk_handle *h = shk_new(k, verbose, min_count, min_qual, chunk_size, do_bloom, do_fit,
                                no_bubble_collapse, no_dead_end_removal);
        if (!h) throw std::runtime_error(shk_new_error_message());
        return AssemblyHelper(h);
    }
    AssemblyHelper(AssemblyHelper &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    AssemblyHelper(const AssemblyHelper &) = delete;
    AssemblyHelper &operator=(const AssemblyHelper &) = delete;
    ~AssemblyHelper() { shk_free(h_); }

    void preprocess(const std::vector<uint8_t> &file1, const std::vector<uint8_t> *file2 = nullptr) {
        check(shk_preprocess(h_, file1.data(), file1.size(), file2 ? file2->data() : nullptr,
                             file2 ? file2->size() : 0));
    }
    std::string get_preprocessing_info() {
        const char *s = shk_get_preprocessing_info(h_);
        if (!s) throw std::runtime_error(shk_last_error(h_));
        return s;
    }
    void assemble() { check(shk_assemble(h_)); }
    std::string get_assembly() {
        const char *s = shk_get_assembly(h_);
        if (!s) throw std::runtime_error(shk_last_error(h_));
        return s;
    }
    void on_state(shk_progress_cb cb, void *user) { shk_set_progress_cb(h_, cb, user); }
    shk_handle *raw() { return h_; }

private:
    explicit AssemblyHelper(shk_handle *h) : h_(h) {}
    void check(int rc) { if (rc != SHK_OK) throw std::runtime_error(shk_last_error(h_)); }
    shk_handle *h_;
};

}  // namespace sparrowhawk
