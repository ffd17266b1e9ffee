// e2e_driver.cpp — TEST INFRASTRUCTURE (oracle/): drives the reference's libllama through its public C API
// (include/llama.h) to measure what scripts/seeddelta-policy-eval.sh:353-469 measures with llama-perplexity
// and llama-cli: windowed perplexity and a greedy continuation, with or without --seeddelta.
//
// Linked twice by oracle/Makefile (target e2e): against libllama_ref.so (the reference's own
// src/llama-seeddelta.cpp) and against libllama_shim.so (llama.cpp_amd/shim/llama-seeddelta-hip.cpp in its
// place; every other object identical). tests/test_gpu_e2e.py runs both on the same GGUF and token file.
//
// Perplexity follows tools/perplexity/perplexity.cpp:441-640 (the default, non-strided form): the token
// stream is cut into windows of n_ctx, the KV cache is cleared per window, the window is decoded in batches
// of n_batch with logits requested for positions >= n_ctx/2, and the negative log-likelihood of token
// [first+j+1] under softmax(logits[first+j]) is accumulated over j < n_ctx-1-first (process_logits /
// log_softmax, :118-170 there: max-subtracted, sum of expf in double). No BOS handling: the synthetic
// GGUFs here use tokenizer.ggml.model = "no_vocab" (src/llama-vocab.cpp:1692), for which add_bos is false.
// Greedy continuation follows llama-cli --temp 0 --top-k 1: argmax of the last position's logits, fed back
// one token at a time.
//
//   sd_e2e_{ref,shim} MODEL.gguf TOKENS.bin --ctx N --batch N --chunks N --threads N [--seeddelta]
//                     [--logits-out F] [--greedy PROMPT_LEN N_PREDICT]
// TOKENS.bin: int32 token ids. Prints one JSON line.
#include "llama.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static void quiet_log(ggml_log_level level, const char * text, void *) {
    if (level == GGML_LOG_LEVEL_ERROR || level == GGML_LOG_LEVEL_WARN) fputs(text, stderr);
}

static double token_nll(const float * logits, int n_vocab, int tok) {
    float mx = logits[0];
    for (int i = 1; i < n_vocab; ++i) mx = std::max(mx, logits[i]);
    double sum = 0.0;
    for (int i = 0; i < n_vocab; ++i) sum += expf(logits[i] - mx);
    return -((double) (logits[tok] - mx) - log(sum));
}

int main(int argc, char ** argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s model.gguf tokens.bin [options]\n", argv[0]); return 2; }
    const char * model_path = argv[1];
    const char * tok_path = argv[2];
    int n_ctx = 64, n_batch = 64, n_chunks = -1, n_threads = 4, greedy_prompt = 0, greedy_n = 0;
    bool seeddelta = false;
    std::string logits_out;
    for (int i = 3; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&]() { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--ctx") n_ctx = atoi(next());
        else if (a == "--batch") n_batch = atoi(next());
        else if (a == "--chunks") n_chunks = atoi(next());
        else if (a == "--threads") n_threads = atoi(next());
        else if (a == "--seeddelta") seeddelta = true;
        else if (a == "--logits-out") logits_out = next();
        else if (a == "--greedy") { greedy_prompt = atoi(next()); greedy_n = atoi(next()); }
        else { fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (n_batch > n_ctx) n_batch = n_ctx;

    std::vector<int32_t> tokens;
    {
        FILE * f = fopen(tok_path, "rb");
        if (!f) { perror(tok_path); return 1; }
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        tokens.resize((size_t) sz / 4);
        if (fread(tokens.data(), 4, tokens.size(), f) != tokens.size()) { fprintf(stderr, "short read\n"); return 1; }
        fclose(f);
    }

    llama_log_set(quiet_log, nullptr);
    llama_backend_init();
    llama_model_params mp = llama_model_default_params();
    mp.n_gpu_layers = 0;
    mp.use_mmap = true;
    llama_model * model = llama_model_load_from_file(model_path, mp);
    if (!model) { fprintf(stderr, "failed to load %s\n", model_path); return 1; }
    llama_context_params cp = llama_context_default_params();
    cp.n_ctx = (uint32_t) std::max(n_ctx, greedy_prompt + greedy_n + 1);
    cp.n_batch = (uint32_t) n_batch;
    cp.n_ubatch = (uint32_t) n_batch;
    cp.n_threads = n_threads;
    cp.n_threads_batch = n_threads;
    cp.seeddelta = seeddelta;
    cp.no_perf = true;
    llama_context * ctx = llama_init_from_model(model, cp);
    if (!ctx) { fprintf(stderr, "failed to create context\n"); return 1; }
    const llama_vocab * vocab = llama_model_get_vocab(model);
    const int n_vocab = llama_vocab_n_tokens(vocab);
    for (int32_t t : tokens) if (t < 0 || t >= n_vocab) { fprintf(stderr, "token %d out of range (%d)\n", t, n_vocab); return 1; }

    // ---- perplexity ------------------------------------------------------------------------------------
    const int n_chunk_max = (int) (tokens.size() / (size_t) n_ctx);
    const int n_chunk = n_chunks < 0 ? n_chunk_max : std::min(n_chunks, n_chunk_max);
    const int first = n_ctx / 2;
    const int num_batches = (n_ctx + n_batch - 1) / n_batch;
    double nll = 0.0;
    int count = 0;
    std::vector<double> chunk_ppl;
    FILE * lf = logits_out.empty() ? nullptr : fopen(logits_out.c_str(), "wb");
    llama_batch batch = llama_batch_init(n_batch, 0, 1);
    std::vector<float> logits;
    for (int i = 0; i < n_chunk; ++i) {
        const int start = i * n_ctx, end = start + n_ctx;
        llama_memory_clear(llama_get_memory(ctx), true);
        logits.clear();
        for (int j = 0; j < num_batches; ++j) {
            const int b0 = start + j * n_batch;
            const int bs = std::min(end - b0, n_batch);
            batch.n_tokens = bs;
            int n_out = 0;
            for (int k = 0; k < bs; ++k) {
                batch.token[k] = tokens[(size_t) (b0 + k)];
                batch.pos[k] = j * n_batch + k;
                batch.n_seq_id[k] = 1;
                batch.seq_id[k][0] = 0;
                batch.logits[k] = batch.pos[k] >= first ? 1 : 0;
                n_out += batch.logits[k] != 0;
            }
            if (llama_decode(ctx, batch)) { fprintf(stderr, "llama_decode failed\n"); return 1; }
            if (n_out > 0) {
                const float * bl = llama_get_logits(ctx);
                logits.insert(logits.end(), bl, bl + (size_t) n_out * n_vocab);
            }
        }
        for (int j = 0; j < n_ctx - 1 - first; ++j)
            nll += token_nll(logits.data() + (size_t) j * n_vocab, n_vocab, tokens[(size_t) (start + first + j + 1)]);
        count += n_ctx - 1 - first;
        chunk_ppl.push_back(exp(nll / count));
        if (lf) fwrite(logits.data(), sizeof(float), logits.size(), lf);
    }
    if (lf) fclose(lf);

    // ---- greedy continuation -----------------------------------------------------------------------------
    std::vector<int32_t> gen;
    if (greedy_n > 0 && greedy_prompt > 0 && (size_t) greedy_prompt <= tokens.size()) {
        llama_memory_clear(llama_get_memory(ctx), true);
        int pos = 0;
        for (int b0 = 0; b0 < greedy_prompt; b0 += n_batch) {
            const int bs = std::min(n_batch, greedy_prompt - b0);
            batch.n_tokens = bs;
            for (int k = 0; k < bs; ++k) {
                batch.token[k] = tokens[(size_t) (b0 + k)];
                batch.pos[k] = pos++;
                batch.n_seq_id[k] = 1;
                batch.seq_id[k][0] = 0;
                batch.logits[k] = (b0 + k == greedy_prompt - 1) ? 1 : 0;
            }
            if (llama_decode(ctx, batch)) { fprintf(stderr, "llama_decode failed\n"); return 1; }
        }
        for (int n = 0; n < greedy_n; ++n) {
            const float * l = llama_get_logits_ith(ctx, -1);
            int best = 0;
            for (int v = 1; v < n_vocab; ++v) if (l[v] > l[best]) best = v;
            gen.push_back(best);
            batch.n_tokens = 1;
            batch.token[0] = best;
            batch.pos[0] = pos++;
            batch.n_seq_id[0] = 1;
            batch.seq_id[0][0] = 0;
            batch.logits[0] = 1;
            if (llama_decode(ctx, batch)) { fprintf(stderr, "llama_decode failed\n"); return 1; }
        }
    }
    llama_batch_free(batch);

    printf("{\"seeddelta\": %s, \"n_ctx\": %d, \"n_batch\": %d, \"chunks\": %d, \"count\": %d, \"nll\": %.17g, \"ppl\": %.17g, \"chunk_ppl\": [",
           seeddelta ? "true" : "false", n_ctx, n_batch, n_chunk, count, nll, count ? exp(nll / count) : 0.0);
    for (size_t i = 0; i < chunk_ppl.size(); ++i) printf("%s%.17g", i ? ", " : "", chunk_ppl[i]);
    printf("], \"greedy\": [");
    for (size_t i = 0; i < gen.size(); ++i) printf("%s%d", i ? ", " : "", gen[i]);
    printf("]}\n");
    fflush(stdout);

    llama_free(ctx);
    llama_model_free(model);
    llama_backend_free();
    return 0;
}
