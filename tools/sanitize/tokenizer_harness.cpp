#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../include/keds_session.h"
void keds_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int main() {
    keds_tokenizer* t = nullptr;
    if (keds_tokenizer_create(getenv("KEDS_BPE_VOCAB"), &t)) return 1;
    std::vector<std::string> texts = {
        "a photo of *", "", " ", "Hello, World!!  \t\n multiple   spaces", "it's John's dog'd 're've'll'm",
        "\xe4\xbd\xa0\xe5\xa5\xbd\xe4\xb8\x96\xe7\x95\x8c", "caf\xc3\xa9 na\xc3\xafve \xc3\x9c\x62\x65r stra\xc3\x9f\x65",
        "&amp;amp; &lt;tag&gt; &#x41; &#65; &nbsp;", "\xf0\x9f\x98\x80\xf0\x9f\x8e\x89 emoji", "\xff\xfe invalid utf8 \xc3",
        "\xce\xa3\xce\x91\xce\xa3 \xce\xa3", "1234567890 12.5% $100 #hash @user", std::string(4000, 'a'),
        std::string("x\0y", 3), "\xe2\x80\x99quote\xe2\x80\x9d", "A\xcc\x8a ngstr\xc3\xb6m \xc4\xb0stanbul",
    };
    for (int rep = 0; rep < 200; ++rep) texts.push_back(std::string(rep % 97 + 1, (char)(33 + rep % 90)) + " " + std::to_string(rep * 7919));
    std::vector<const char*> ptrs;
    for (auto& s : texts) ptrs.push_back(s.c_str());
    std::vector<int32_t> out(texts.size() * 77);
    int rc = keds_tokenize(t, ptrs.data(), (int)ptrs.size(), 77, 1, out.data());
    long sum = 0; for (int v : out) sum += v;
    printf("rc=%d checksum=%ld\n", rc, sum);
    std::vector<int32_t> out2(77);
    const char* longt = ptrs[12];
    rc = keds_tokenize(t, &longt, 1, 77, 0, out2.data());   // too long without truncate: error path
    printf("rc(no truncate)=%d\n", rc);
    keds_tokenizer_destroy(t);
    return 0;
}
