// TEST INFRASTRUCTURE (oracle side, never linked into the product).
//
// Runs the reference's OWN rank-pair generator -- PairwiseRankGenerator, apex_svd_data.cpp:812-1025, obtained the way
// svd_feature.cpp:128-143 obtains it: create_plus_iterator(input_type::BINARY_BUFFER_RANK) + set_param + init --
// over a user-group buffer file and writes the blocks it produces, for `rounds` passes, as one user-group buffer file
// (SVDPlusBlock::save_to_file, apex_svd_data.h:419-431).  Built by oracle/Makefile from the reference sources where
// they lie into oracle/_ref/ref_pairgen_dump; tests compare the product's sampler with its output.
//
//   ref_pairgen_dump <in.buffer> <out.buffer> <seed> <rounds> [name=value ...]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "apex_svd_data.h"

int main(int argc, char **argv) {
    if (argc < 5) {
        fprintf(stderr, "usage: %s <in.buffer> <out.buffer> <seed> <rounds> [name=value ...]\n", argv[0]);
        return 2;
    }
    using namespace apex_svd;
    srand((unsigned)atoi(argv[3]));   // svd_feature.cpp:293 seeds the same libc stream
    const int rounds = atoi(argv[4]);
    IDataIterator<SVDPlusBlock> *itr = create_plus_iterator(input_type::BINARY_BUFFER_RANK);
    itr->set_param("buffer_feature", argv[1]);
    for (int i = 5; i < argc; i++) {
        std::string kv(argv[i]);
        size_t eq = kv.find('=');
        if (eq == std::string::npos) { fprintf(stderr, "bad argument %s\n", argv[i]); return 2; }
        itr->set_param(kv.substr(0, eq).c_str(), kv.substr(eq + 1).c_str());
    }
    itr->init();
    FILE *fo = fopen(argv[2], "wb");
    if (!fo) { perror(argv[2]); return 1; }
    int head[4] = {0, 0, 0, 0};
    fwrite(head, sizeof(int), 4, fo);
    SVDPlusBlock e;
    for (int r = 0; r < rounds; r++) {
        while (itr->next(e)) {
            e.save_to_file(fo);
            head[0]++;
            if (e.num_ufeedback > head[1]) head[1] = e.num_ufeedback;
            if (e.data.num_row > head[2]) head[2] = e.data.num_row;
            if (e.data.num_val > head[3]) head[3] = e.data.num_val;
        }
        itr->before_first();
    }
    fseek(fo, 0, SEEK_SET);
    fwrite(head, sizeof(int), 4, fo);
    fclose(fo);
    delete itr;
    return 0;
}
