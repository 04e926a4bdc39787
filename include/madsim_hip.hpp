// madsim_hip.hpp — C++17 host-side mirror of madsim's seed driver over the C-ABI (include/madsim_hip.h).
//
// The reference's host code is Rust; this image has no Rust toolchain, so the host side above the C-ABI
// is written in C++ with the reference's own names and behaviour:
//   madsim::runtime::Builder            madsim/src/sim/runtime/builder.rs:7-22   (same public fields)
//   Builder::from_env()                 builder.rs:64-118                         (same environment variables)
//   Builder::run(workload)              builder.rs:121-162                        (returns on success; on the
//                                       first failing seed prints the reproduction note of
//                                       runtime/mod.rs:205-210 and throws — the C++ stand-in for the panic)
//   madsim::WorkloadBuilder / Task      the body of a #[madsim::test] as an actor program; method names are
//                                       the reference API calls they stand for (net/endpoint.rs, time/sleep.rs,
//                                       task/mod.rs, runtime/mod.rs:276-303, net/mod.rs:164-222)
// Header-only; link with libmadsim_hip.so.  There is no CPU fallback: without a GPU every run throws.
#ifndef MADSIM_HIP_HPP
#define MADSIM_HIP_HPP

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "madsim_hip.h"

namespace madsim {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// A failing seed: what `cargo test` would report as the test's panic.
struct SimulationFailure : std::runtime_error {
    uint64_t seed;
    madsim_result_t result;
    SimulationFailure(uint64_t s, const madsim_result_t& r)
        : std::runtime_error("seed " + std::to_string(s) + ": " + verdict_name(r.verdict)), seed(s), result(r) {}
    static const char* verdict_name(uint32_t v) {
        static const char* n[] = {"pass", "panic", "no events, all tasks will block forever", "time limit exceeded",
                                  "device capacity overflow", "step limit", "outside the workload model", "internal invariant"};
        return v < 8 ? n[v] : "?";
    }
};

inline void check(int rc) {
    if (rc < 0) throw Error(rc, std::string(madsim_hip_strerror(rc)) + ": " + madsim_hip_last_error());
}

// madsim::Config.net (net/network.rs:66-89)
struct Config {
    double packet_loss_rate = 0.0;
    uint64_t send_latency_start_ns = 1000000, send_latency_end_ns = 10000000;
    bool buggify = false;
    // ranges Task::set_latency(i) switches to — NetSim::update_config(|c| c.send_latency = ..) (net/mod.rs:138-141); at most four {start, end} in ns
    std::vector<std::pair<uint64_t, uint64_t>> latency_table;

    // `content.parse::<Config>()` of MADSIM_TEST_CONFIG (builder.rs:81-88; config.rs:29-35 = toml::from_str): the [net] table —
    // packet_loss_rate and send_latency = { start = { secs, nanos }, end = { secs, nanos } } — in either TOML spelling (inline
    // tables as in config.rs:52-56, or [net.send_latency.start] sections as Display prints them); [tcp] is an empty struct.
    // A TOML subset on purpose: tables, inline tables, numbers, strings and booleans — what a Config file can hold.
    static Config from_toml(const std::string& text) {
        Config c;
        std::optional<uint64_t> s_secs, s_nanos, e_secs, e_nanos;
        auto set = [&](const std::string& path, const std::string& v) {
            auto num = [&]() -> double {
                char* end = nullptr; double x = std::strtod(v.c_str(), &end);
                if (v.empty() || *end) throw std::invalid_argument("failed to parse config file: `" + path + "` is not a number");
                return x;
            };
            if (path == "net.packet_loss_rate") c.packet_loss_rate = num();
            else if (path == "net.send_latency.start.secs") s_secs = (uint64_t)num();
            else if (path == "net.send_latency.start.nanos") s_nanos = (uint64_t)num();
            else if (path == "net.send_latency.end.secs") e_secs = (uint64_t)num();
            else if (path == "net.send_latency.end.nanos") e_nanos = (uint64_t)num();
            else if (path.rfind("net.", 0) == 0) throw std::invalid_argument("failed to parse config file: unknown field `" + path + "`");
            // (other tables — [tcp] — carry no fields this path reads)
        };
        size_t i = 0; const size_t n = text.size();
        auto ws = [&](bool newlines) {
            while (i < n) {
                if (text[i] == '#') { while (i < n && text[i] != '\n') i++; }
                else if (text[i] == ' ' || text[i] == '\t' || text[i] == '\r' || (newlines && text[i] == '\n')) i++;
                else break;
            }
        };
        auto key = [&]() -> std::string {              // bare or quoted key, dotted: a.b."c"
            std::string k;
            for (;;) {
                ws(false);
                if (i < n && (text[i] == '"' || text[i] == '\'')) { const char q = text[i++]; while (i < n && text[i] != q) k += text[i++]; if (i < n) i++; }
                else { const size_t b = i; while (i < n && (std::isalnum((unsigned char)text[i]) || text[i] == '_' || text[i] == '-')) i++; if (i == b) throw std::invalid_argument("failed to parse config file: key expected"); k.append(text, b, i - b); }
                ws(false);
                if (i < n && text[i] == '.') { k += '.'; i++; } else return k;
            }
        };
        std::function<void(const std::string&)> value = [&](const std::string& path) {
            ws(false);
            if (i < n && text[i] == '{') {               // inline table
                i++; ws(true);
                while (i < n && text[i] != '}') {
                    const std::string k = key();
                    if (i >= n || text[i] != '=') throw std::invalid_argument("failed to parse config file: `=` expected");
                    i++; value(path + "." + k); ws(true);
                    if (i < n && text[i] == ',') { i++; ws(true); }
                }
                if (i >= n) throw std::invalid_argument("failed to parse config file: unterminated inline table");
                i++;
            } else if (i < n && (text[i] == '"' || text[i] == '\'')) {
                const char q = text[i++]; std::string v; while (i < n && text[i] != q) v += text[i++]; if (i < n) i++;
                set(path, v);
            } else {
                const size_t b = i;
                while (i < n && text[i] != ',' && text[i] != '}' && text[i] != '\n' && text[i] != '#' && text[i] != ' ' && text[i] != '\t' && text[i] != '\r') i++;
                std::string v(text, b, i - b);
                v.erase(std::remove(v.begin(), v.end(), '_'), v.end());       // 1_000_000
                set(path, v);
            }
        };
        std::string table;
        for (ws(true); i < n; ws(true)) {
            if (text[i] == '[') {
                i++; table = key();
                if (i >= n || text[i] != ']') throw std::invalid_argument("failed to parse config file: `]` expected");
                i++;
            } else {
                const std::string k = key();
                if (i >= n || text[i] != '=') throw std::invalid_argument("failed to parse config file: `=` expected");
                i++; value(table.empty() ? k : table + "." + k);
            }
        }
        if (s_secs || s_nanos || e_secs || e_nanos) {
            if (!(s_secs && s_nanos && e_secs && e_nanos)) throw std::invalid_argument("failed to parse config file: send_latency needs start and end, secs and nanos");
            c.send_latency_start_ns = *s_secs * 1000000000ull + *s_nanos; c.send_latency_end_ns = *e_secs * 1000000000ull + *e_nanos;
        }
        return c;
    }
    madsim_config_t raw() const {
        madsim_config_t c{};
        c.packet_loss_rate = packet_loss_rate; c.lat_lo_ns = send_latency_start_ns; c.lat_hi_ns = send_latency_end_ns;
        c.buggify = buggify ? 1 : 0;
        if (latency_table.size() > 4) throw std::invalid_argument("at most four latency_table entries");
        c.n_lat_table = (uint32_t)latency_table.size();
        for (size_t i = 0; i < latency_table.size(); i++) { c.lat_table_lo_ns[i] = latency_table[i].first; c.lat_table_hi_ns[i] = latency_table[i].second; }
        return c;
    }
};

class WorkloadBuilder;

// One async block (`node.spawn(async move { ... })`).
class Task {
  public:
    int index() const { return index_; }
    int label() const { return (int)code_.size(); }
    Task& done() { return emit(MS_OP_DONE); }
    Task& spawn(const Task& t) { return emit(MS_OP_SPAWN, (uint8_t)t.index_); }
    Task& join(const Task& t, bool expect_err = false) { return emit(MS_OP_JOIN, (uint8_t)t.index_, expect_err ? 1 : 0); }
    Task& yield_now() { return emit(MS_OP_YIELD); }
    Task& panic(uint8_t code = 0) { return emit(MS_OP_PANIC, 0, 0, code); }            // panic!("<code>"): the message is the decimal text of `code`
    Task& panic(const std::string& message);                                           // panic!("<message>"): interned by WorkloadBuilder::build()
    Task& panic_with_flag(int flag, int32_t offset = 0) { return emit(MS_OP_PANIC, 1, (uint16_t)flag, (uint32_t)offset); }   // panic!("{}", flag + offset)
    Task& set(int reg, uint32_t v) { return emit(MS_OP_SET, (uint8_t)reg, 0, v); }
    Task& djnz(int reg, int target) { return emit(MS_OP_DJNZ, (uint8_t)reg, (uint16_t)target, 0, true); }
    Task& jmp(int target) { return emit(MS_OP_JMP, 0, (uint16_t)target, 0, true); }
    Task& trace(uint32_t v) { return emit(MS_OP_TRACE, 0, 0, v); }
    Task& sleep(std::chrono::nanoseconds d) { return dur(MS_OP_SLEEP, 0, d); }
    Task& mark() { return emit(MS_OP_MARK); }
    Task& sleep_until(std::chrono::nanoseconds after_mark) { return dur(MS_OP_SLEEP_UNTIL, 0, after_mark); }
    Task& assert_elapsed_eq(std::chrono::nanoseconds d) { return dur(MS_OP_ASSERT_ELAPSED, 0, d); }
    Task& assert_elapsed_ge(std::chrono::nanoseconds d) { return dur(MS_OP_ASSERT_ELAPSED, 1, d); }
    Task& bind(int addr, bool port_to_val = false) { return emit(MS_OP_BIND, (uint8_t)addr, port_to_val ? 2 : 0); }   // val = local_addr().port()
    Task& try_bind(int addr) { return emit(MS_OP_BIND, (uint8_t)addr, 1); }            // val = 0 | MADSIM_VAL_ADDR_NOT_AVAILABLE | MADSIM_VAL_ADDR_IN_USE
    Task& send_to(int ep, int dst, uint8_t tag, uint32_t payload) { return emit(MS_OP_SEND, (uint8_t)ep, (uint16_t)((tag << 8) | dst), payload); }
    Task& reply(int ep, uint8_t tag, uint32_t payload) { return emit(MS_OP_REPLY, (uint8_t)ep, (uint16_t)(tag << 8), payload); }
    Task& recv_from(int ep, uint8_t tag) { return emit(MS_OP_RECV, (uint8_t)ep, (uint16_t)(tag << 8)); }
    Task& assert_val(uint32_t v) { return emit(MS_OP_ASSERT_VAL, 0, 0, v); }
    Task& close(int ep) { return emit(MS_OP_CLOSE, (uint8_t)ep); }
    // timeouts / random sleeps / branching on the received value
    Task& recv_from_timeout(int ep, uint8_t tag, std::chrono::nanoseconds d) {
        uint64_t ns = (uint64_t)d.count();
        return emit(MS_OP_RECV_TIMEOUT, (uint8_t)ep, (uint16_t)((tag << 8) | (ns / 1000000000ull)), (uint32_t)(ns % 1000000000ull));
    }
    Task& sleep_rand(std::chrono::milliseconds lo /* multiple of 50 ms */, std::chrono::nanoseconds hi) {
        uint64_t ns = (uint64_t)hi.count();
        return emit(MS_OP_SLEEP_RAND, (uint8_t)(lo.count() / 50), (uint16_t)(ns / 1000000000ull), (uint32_t)(ns % 1000000000ull));
    }
    Task& random_u32() { return emit(MS_OP_RANDOM, 0); }
    Task& getrandom_byte() { return emit(MS_OP_RANDOM, 1); }
    Task& trace_system_time() { return emit(MS_OP_TRACE_TIME, 0); }
    Task& trace_instant() { return emit(MS_OP_TRACE_TIME, 1); }
    Task& trace_val() { return emit(MS_OP_TRACE_TIME, 2); }
    Task& rand_bool(int table_index) { return emit(MS_OP_RAND_BOOL, (uint8_t)table_index); }
    Task& jeq(uint32_t value, int target) { return emit(MS_OP_JEQ, 0, (uint16_t)target, value, true); }
    // reliable channel (Endpoint::connect1 / accept1)
    Task& connect1(int ep, int dst) { return emit(MS_OP_CONNECT, (uint8_t)ep, (uint16_t)dst); }
    Task& accept1(int ep) { return emit(MS_OP_ACCEPT, (uint8_t)ep); }
    Task& chan_send(uint32_t payload) { return emit(MS_OP_CSEND, 0, 0, payload); }
    Task& chan_recv() { return emit(MS_OP_CRECV); }
    Task& chan_close() { return emit(MS_OP_CCLOSE); }
    Task& spawn_move_conn(const Task& t) { return emit(MS_OP_SPAWN, (uint8_t)t.index_, MADSIM_SPAWN_MOVE_CONN); }
    // typed RPC (Endpoint::call / call_timeout / add_rpc_handler, net/rpc.rs:96-180): req_id = R::ID - 0x80, 8-bit codes
    Task& rpc_call(int ep, int dst, uint8_t req_id, uint8_t code, std::chrono::milliseconds timeout = std::chrono::milliseconds(0)) {
        return emit(MS_OP_RPC_CALL, (uint8_t)ep, (uint16_t)(((MADSIM_TAG_RPC_FIRST + req_id) << 8) | dst), ((uint32_t)timeout.count() << 8) | code);
    }
    Task& rpc_recv(int ep, uint8_t req_id) { return emit(MS_OP_RECV, (uint8_t)ep, (uint16_t)((MADSIM_TAG_RPC_FIRST + req_id) << 8)); }
    Task& rpc_reply(int ep, uint8_t code) { return emit(MS_OP_RPC_REPLY, (uint8_t)ep, 0, code); }
    // NetSim::hook_rpc_req / hook_rpc_rsp (net/mod.rs:240-284): drop requests R leaving `node` / responses on their way to `node`
    Task& hook_rpc_req(int node, uint8_t req_id, std::optional<uint8_t> code = std::nullopt) {
        return emit(MS_OP_HOOK_REQ, (uint8_t)node, (uint16_t)(((MADSIM_TAG_RPC_FIRST + req_id) << 8) | (code ? 0 : 1)), code ? *code : 0);
    }
    Task& hook_rpc_rsp(int node, std::optional<uint8_t> code = std::nullopt) { return emit(MS_OP_HOOK_RSP, (uint8_t)node, code ? 0 : 1, code ? *code : 0); }
    // NetSim::global_ipvs() at run time (net/ipvs.rs:50-85); `service` = the index ipvs_service returned
    Task& ipvs_add_service(int service) { return emit(MS_OP_IPVS, MADSIM_IPVS_ADD_SERVICE, (uint16_t)service); }
    Task& ipvs_del_service(int service) { return emit(MS_OP_IPVS, MADSIM_IPVS_DEL_SERVICE, (uint16_t)service); }
    Task& ipvs_add_server(int service, int server) { return emit(MS_OP_IPVS, MADSIM_IPVS_ADD_SERVER, (uint16_t)service, (uint32_t)server); }
    Task& ipvs_del_server(int service, int server) { return emit(MS_OP_IPVS, MADSIM_IPVS_DEL_SERVER, (uint16_t)service, (uint32_t)server); }
    Task& spawn_move_request(const Task& t) { return emit(MS_OP_SPAWN, (uint8_t)t.index_, MADSIM_SPAWN_MOVE_REQUEST); }
    // supervisor (Handle::kill / restart / pause / resume / is_exit, JoinHandle::abort)
    Task& kill(int node) { return emit(MS_OP_KILL, (uint8_t)node); }
    Task& restart(int node) { return emit(MS_OP_RESTART, (uint8_t)node); }
    Task& pause(int node) { return emit(MS_OP_PAUSE, (uint8_t)node); }
    Task& resume(int node) { return emit(MS_OP_RESUME, (uint8_t)node); }
    Task& abort(const Task& t) { return emit(MS_OP_ABORT, (uint8_t)t.index_); }
    Task& assert_exit(int node, bool expected) { return emit(MS_OP_ASSERT_EXIT, (uint8_t)node, expected ? 1 : 0); }
    Task& build_node(int node) { return emit(MS_OP_BUILD, (uint8_t)node); }
    // shared Arc<AtomicUsize>-style flags
    Task& flag_store(int flag, uint32_t v) { return emit(MS_OP_GSET, (uint8_t)flag, 0, v); }
    Task& flag_add(int flag, uint32_t v) { return emit(MS_OP_GADD, (uint8_t)flag, 0, v); }
    Task& assert_flag(int flag, uint32_t v) { return emit(MS_OP_ASSERT_G, (uint8_t)flag, 0, v); }
    Task& panic_if_flag_lt(int flag, uint32_t v) { return emit(MS_OP_PANIC_IF_G_LT, (uint8_t)flag, 0, v); }
    Task& clog_node(int node) { return emit(MS_OP_CLOG_NODE, (uint8_t)node, 3); }
    Task& unclog_node(int node) { return emit(MS_OP_UNCLOG_NODE, (uint8_t)node, 3); }
    Task& set_latency(int index) { return emit(MS_OP_SET_LATENCY, (uint8_t)index); }   // NetSim::update_config(|c| c.send_latency = latency_table[index])
    Task& clog_link(int src, int dst) { return emit(MS_OP_CLOG_LINK, (uint8_t)src, (uint16_t)dst); }
    Task& unclog_link(int src, int dst) { return emit(MS_OP_UNCLOG_LINK, (uint8_t)src, (uint16_t)dst); }

  private:
    friend class WorkloadBuilder;
    struct Ins { madsim_insn_t in; bool reloc; std::string text; };     // text: the literal message of a panic("..")
    Task(int index, int node, uint8_t flags) : index_(index), node_(node), flags_(flags) {}
    Task& emit(uint8_t op, uint8_t a = 0, uint16_t b = 0, uint32_t imm = 0, bool reloc = false) {
        code_.push_back({madsim_insn_t{op, a, b, imm}, reloc, std::string()});
        return *this;
    }
    Task& dur(uint8_t op, uint8_t a, std::chrono::nanoseconds d) {
        uint64_t ns = (uint64_t)d.count();
        return emit(op, a, (uint16_t)(ns / 1000000000ull), (uint32_t)(ns % 1000000000ull));
    }
    int index_, node_;
    uint8_t flags_;
    std::vector<Ins> code_;
};

// Owns the tables a madsim_workload_t points into.
struct Workload {
    std::vector<madsim_node_t> nodes;
    std::vector<madsim_prog_t> progs;
    std::vector<madsim_sock_t> socks;
    std::vector<madsim_insn_t> insns;
    std::vector<madsim_service_t> services;   // IPVS virtual services (net/ipvs.rs)
    std::vector<uint32_t> panic_match;        // empty, or 8 words per node: which message codes restart it (madsim_workload_t.panic_match)
    uint32_t panic_dyn_max = 0;
    madsim_workload_t raw() const {
        madsim_workload_t w{};
        w.n_nodes = (uint32_t)nodes.size() - 1; w.n_progs = (uint32_t)progs.size(); w.n_socks = (uint32_t)socks.size();
        w.n_insns = (uint32_t)insns.size(); w.nodes = nodes.data(); w.progs = progs.data(); w.socks = socks.data(); w.insns = insns.data();
        w.n_services = (uint32_t)services.size(); w.services = services.empty() ? nullptr : services.data();
        w.panic_dyn_max = panic_dyn_max; w.panic_match = panic_match.empty() ? nullptr : panic_match.data();
        return w;
    }
};

class WorkloadBuilder {
  public:
    WorkloadBuilder() { tasks_.reserve(256); nodes_.push_back(madsim_node_t{}); tasks_.push_back(Task(0, 0, 0)); }   // Task& stay valid
    Task& main() { return tasks_[0]; }                                   // the future handed to block_on
    // Byte strings on the wire never steer the simulation (a Payload is a Box<dyn Any>, endpoint.rs:69-94): the test body can
    // only compare them, so they are interned — equal bytes <=> equal value.  payload(): the 32-bit value of a datagram /
    // channel payload; rpc_message(): the 8-bit code of a typed-RPC message together with its call_with_data bytes
    // (net/rpc.rs:114-131), at most 256 distinct pairs per workload.
    uint32_t payload(const std::string& data) {
        for (size_t i = 0; i < payloads_.size(); i++) if (payloads_[i] == data) return 0x40000000u + (uint32_t)i;
        payloads_.push_back(data); return 0x40000000u + (uint32_t)payloads_.size() - 1;
    }
    uint8_t rpc_message(const std::string& msg, const std::string& data = std::string()) {
        const std::string key = std::to_string(msg.size()) + ":" + msg + data;
        for (size_t i = 0; i < rpc_messages_.size(); i++) if (rpc_messages_[i] == key) return (uint8_t)i;
        if (rpc_messages_.size() >= 256) throw std::length_error("at most 256 distinct typed-RPC (message, data) pairs");
        rpc_messages_.push_back(key); return (uint8_t)(rpc_messages_.size() - 1);
    }
    // Handle::create_node()[.ip(10.0.0.<id>)][.restart_on_panic()][.restart_on_panic_matching(code)..].build()
    int create_node(bool restart_on_panic = false, std::vector<uint8_t> restart_on_panic_matching = {}, bool ip = true) {
        if (restart_on_panic_matching.size() > 2) throw std::length_error("at most two restart_on_panic_matching patterns");
        madsim_node_t n{};
        n.flags = (uint8_t)((restart_on_panic ? MADSIM_NODE_RESTART_ON_PANIC : 0) | (ip ? 0 : MADSIM_NODE_NO_IP) |
                            (restart_on_panic_matching.empty() ? 0 : MADSIM_NODE_RESTART_MATCHING));
        n.n_match = (uint8_t)restart_on_panic_matching.size();
        for (size_t i = 0; i < restart_on_panic_matching.size(); i++) n.match[i] = restart_on_panic_matching[i];
        nodes_.push_back(n); return (int)nodes_.size() - 1;
    }
    // ...restart_on_panic_matching("pattern")...: substrings of the panic message (`error_msg.contains(s)`, task/mod.rs:297-300),
    // any number of them; build() evaluates them against every message code (literal messages, decimal texts of numbers)
    int create_node_matching(std::vector<std::string> patterns, bool ip = true) {
        madsim_node_t n{};
        n.flags = (uint8_t)(MADSIM_NODE_RESTART_MATCHING | (ip ? 0 : MADSIM_NODE_NO_IP));
        nodes_.push_back(n);
        patterns_.resize(nodes_.size());
        patterns_.back() = std::move(patterns);
        return (int)nodes_.size() - 1;
    }
    // a virtual service address that belongs to no node ("1.1.1.<ip_id>:port"): a destination only
    int virtual_addr(uint8_t ip_id, uint16_t port) { socks_.push_back(madsim_sock_t{ip_id, MADSIM_ADDR_VIRTUAL, port}); return (int)socks_.size() - 1; }
    // ipvs.add_service(ServiceAddr::Tcp(vaddr), RoundRobin) + add_server per entry (net/ipvs.rs:50-85), before any task runs
    // (absent = only the address is declared: the service exists once a task calls ipvs_add_service).  Returns the service index.
    int ipvs_service(int vaddr, const std::vector<int>& servers = {}, bool absent = false) {
        if (services_.size() >= MADSIM_MAX_SERVICES || servers.size() > 6 || (absent && !servers.empty())) throw std::length_error("at most 8 services of at most 6 servers");
        madsim_service_t s{};
        s.vaddr = (uint8_t)vaddr; s.n_servers = absent ? (uint8_t)MADSIM_SERVICE_ABSENT : (uint8_t)servers.size();
        for (size_t i = 0; i < servers.size(); i++) s.servers[i] = (uint8_t)servers[i];
        services_.push_back(s);
        return (int)services_.size() - 1;
    }
    // 10.0.0.<node>:port, or 0.0.0.0:port / 127.0.0.1:port as used on `node` (kind = MADSIM_ADDR_*).  port 0 = an ephemeral
    // Endpoint (network.rs:224-236): each bind gets the node's lowest free port for that IP; not a destination operand.
    int addr(int node, uint16_t port, uint8_t kind = MADSIM_ADDR_IP) {
        socks_.push_back(madsim_sock_t{(uint8_t)node, kind, port}); return (int)socks_.size() - 1;
    }
    // spawn_on_drop: the body owns a guard whose Drop calls task::spawn(<the NEXT task declared>) (MADSIM_PROG_DROP_SPAWN)
    Task& task(int node, bool init = false, bool before_block_on = false, bool spawn_on_drop = false) {
        if (tasks_.size() >= 255) throw std::length_error("at most 255 task programs");
        tasks_.push_back(Task((int)tasks_.size(), node, (uint8_t)((init ? MADSIM_PROG_INIT : 0) | (before_block_on ? MADSIM_PROG_PRE : 0) |
                                                                 (spawn_on_drop ? MADSIM_PROG_DROP_SPAWN : 0))));
        return tasks_.back();
    }
    Workload build() {
        Workload w;
        w.nodes = nodes_; w.socks = socks_; w.services = services_;
        // message codes: literal messages are interned from 254 downwards, a number is its decimal text; a node's row has bit c
        // set when one of its patterns is a substring of the text of code c (the rule of madsim_amd/workload.py::_panic_rows)
        std::vector<std::string> literals;
        for (auto& t : tasks_) for (auto& i : t.code_) if (!i.text.empty() && std::find(literals.begin(), literals.end(), i.text) == literals.end()) literals.push_back(i.text);
        std::sort(literals.begin(), literals.end());
        if (literals.size() > 200) throw std::length_error("at most 200 distinct literal panic messages");
        const uint32_t dyn_max = 254 - (uint32_t)literals.size();
        auto text_of = [&](uint32_t c) { return c > dyn_max ? literals[254 - c] : std::to_string(c); };
        bool any_patterns = false;
        for (auto& p : patterns_) any_patterns |= !p.empty();
        if (any_patterns || !literals.empty()) {
            w.panic_dyn_max = dyn_max;
            w.panic_match.assign(8 * nodes_.size(), 0);
            for (size_t n = 0; n < patterns_.size(); n++)
                for (uint32_t c = 0; c <= 254; c++)
                    for (auto& p : patterns_[n]) if (text_of(c).find(p) != std::string::npos) w.panic_match[8 * n + (c >> 5)] |= 1u << (c & 31);
            for (size_t n = 0; n < nodes_.size(); n++)          // numeric patterns given to create_node(): their decimal text
                for (uint32_t k = 0; k < nodes_[n].n_match && k < 2; k++)
                    for (uint32_t c = 0; c <= 254; c++)
                        if (text_of(c).find(std::to_string(nodes_[n].match[k])) != std::string::npos) w.panic_match[8 * n + (c >> 5)] |= 1u << (c & 31);
        }
        for (auto& t : tasks_) {
            uint16_t base = (uint16_t)w.insns.size();
            if (t.code_.empty() || (t.code_.back().in.op != MS_OP_DONE && t.code_.back().in.op != MS_OP_JMP)) t.done();
            w.progs.push_back(madsim_prog_t{(uint8_t)t.node_, t.flags_, base});
            for (auto& i : t.code_) {
                madsim_insn_t in = i.in;
                if (!i.text.empty()) in.imm = 254 - (uint32_t)(std::find(literals.begin(), literals.end(), i.text) - literals.begin());
                else if (in.op == MS_OP_PANIC && in.a == 0 && in.imm > dyn_max) throw std::invalid_argument("numeric panic code collides with a literal message");
                if (i.reloc) in.b = (uint16_t)(in.b + base);
                w.insns.push_back(in);
            }
        }
        return w;
    }

  private:
    std::vector<madsim_node_t> nodes_;
    std::vector<madsim_sock_t> socks_;
    std::vector<Task> tasks_;
    std::vector<std::string> payloads_, rpc_messages_;
    std::vector<std::vector<std::string>> patterns_;      // per node: restart_on_panic_matching strings
    std::vector<madsim_service_t> services_;
};

inline Task& Task::panic(const std::string& message) {
    if (message.empty()) throw std::invalid_argument("empty panic message");
    code_.push_back({madsim_insn_t{MS_OP_PANIC, 0, 0, 0}, false, message});
    return *this;
}

namespace runtime {

// runtime/mod.rs:205-210
inline void panic_with_info(uint64_t seed) {
    std::fprintf(stderr, "note: run with `MADSIM_TEST_SEED=%llu` environment variable to reproduce this error\n",
                 (unsigned long long)seed);
}

struct Builder {
    uint64_t seed = 0;
    uint64_t count = 1;
    uint16_t jobs = 1;
    Config config;
    std::optional<double> time_limit;        // seconds
    bool check = false;
    bool allow_system_thread = false;
    int device = 0;
    madsim_limits_t capacities{};            // device capacities to start from (no reference counterpart; 0 = defaults);
                                             // capacities.no_trace_hash = 1: results without the determinism-log fingerprint (4 % faster on ping-pong)

    // builder.rs:64-118
    static Builder from_env() {
        Builder b;
        auto env = [](const char* k) -> const char* { return std::getenv(k); };
        auto parse_u64 = [](const char* s, const char* what) -> uint64_t {
            char* end = nullptr;
            unsigned long long v = std::strtoull(s, &end, 10);
            if (!s[0] || *end) throw std::invalid_argument(std::string(what) + " should be an integer");
            return v;
        };
        if (auto s = env("MADSIM_TEST_SEED")) b.seed = parse_u64(s, "MADSIM_TEST_SEED");
        else b.seed = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
                          std::chrono::system_clock::now().time_since_epoch()).count();
        if (auto s = env("MADSIM_TEST_JOBS")) b.jobs = (uint16_t)parse_u64(s, "MADSIM_TEST_JOBS");
        if (auto s = env("MADSIM_TEST_CONFIG")) {
            std::ifstream f(s);
            if (!f) throw std::invalid_argument("failed to read config file");
            std::stringstream ss; ss << f.rdbuf();
            b.config = Config::from_toml(ss.str());
        }
        if (auto s = env("MADSIM_TEST_NUM")) b.count = parse_u64(s, "MADSIM_TEST_NUM");
        if (auto s = env("MADSIM_TEST_TIME_LIMIT")) {
            char* end = nullptr;
            double v = std::strtod(s, &end);
            if (!s[0] || *end) throw std::invalid_argument("MADSIM_TEST_TIME_LIMIT should be an number");
            b.time_limit = v;
        }
        b.check = env("MADSIM_TEST_CHECK_DETERMINISM") != nullptr;
        if (b.check && b.count < 2) b.count = 2;
        b.allow_system_thread = env("MADSIM_ALLOW_SYSTEM_THREAD") != nullptr;
        return b;
    }

    // Seed search: seed .. seed + count as batches the LIBRARY keeps in flight on its own streams (madsim_hip_run_campaign),
    // stopping at the first batch that holds a failing seed; prints the reproduction note for the seed it found.  No per-seed
    // results: re-run the reported seed for details.
    madsim_campaign_t search_first_failure(const Workload& wl) const {
        madsim::check(madsim_hip_init(device));
        madsim_workload_t w = wl.raw();
        madsim_config_t cfg = config.raw();
        madsim_limits_t lim = capacities;
        if (time_limit) { lim.time_limit_ns = (uint64_t)(*time_limit * 1e9 + 0.5); if (!lim.time_limit_ns) lim.time_limit_ns = 1; }
        madsim_campaign_t rep{};
        madsim::check(madsim_hip_run_campaign(&w, &cfg, seed, count, 0, 0, MADSIM_CAMPAIGN_STOP_AT_FAILURE, &lim, &rep));
        if (rep.first_failing_seed != UINT64_MAX) panic_with_info(rep.first_failing_seed);
        return rep;
    }

    // builder.rs:121-162: run seeds seed..seed+count; return on success, "panic" on the first failing seed.
    // Reports the numerically smallest failing seed (the reference reports the first to complete).
    std::vector<madsim_result_t> run(const Workload& wl) const {
        madsim::check(madsim_hip_init(device));
        madsim_workload_t w = wl.raw();
        madsim_config_t cfg = config.raw();
        madsim_limits_t lim = capacities;
        // Some(Duration::ZERO) is a limit too (panics at the first idle advance); 0 means None in the C-ABI
        if (time_limit) { lim.time_limit_ns = (uint64_t)(*time_limit * 1e9 + 0.5); if (!lim.time_limit_ns) lim.time_limit_ns = 1; }
        if (check) {                                   // Runtime::check_determinism (runtime/mod.rs:178-202)
            std::vector<uint8_t> l1(1 << 20), l2(1 << 20);
            madsim_result_t r1{}, r2{};
            madsim_limits_t nolim = capacities;           // check_determinism never sets a time limit (runtime/mod.rs:178-202)
            int64_t n1 = madsim_hip_trace_seed(&w, &cfg, seed, &nolim, l1.data(), l1.size(), &r1);
            int64_t n2 = madsim_hip_trace_seed(&w, &cfg, seed, &nolim, l2.data(), l2.size(), &r2);
            if (n1 < 0) madsim::check((int)n1);
            if (n2 < 0) madsim::check((int)n2);
            size_t n = (size_t)(n1 < (int64_t)l1.size() ? n1 : (int64_t)l1.size());
            if (n1 != n2 || std::memcmp(l1.data(), l2.data(), n) != 0) {
                panic_with_info(seed);
                throw std::runtime_error("non-determinism detected");
            }
            if (r1.verdict != MADSIM_PASS) { panic_with_info(seed); throw SimulationFailure(seed, r1); }
            return {r1};
        }
        std::vector<madsim_result_t> out(count);
        madsim_summary_t s{};
        madsim::check(madsim_hip_run_batch_auto(&w, &cfg, seed, count, &lim, out.data(), &s, 6));   // runner verdicts are re-run
        if (s.n_failed) {
            auto runner = [](uint32_t v) { return MADSIM_IS_RUNNER_VERDICT(v); };
            // a genuine test failure wins over unresolved runner limits: the first failing seed and its note are never hidden
            for (uint64_t i = 0; i < count; i++)
                if (out[i].verdict != MADSIM_PASS && !runner(out[i].verdict)) { panic_with_info(seed + i); throw SimulationFailure(seed + i, out[i]); }
            uint64_t i = 0;
            while (!runner(out[i].verdict)) i++;
            // a capacity / step-cap verdict that survived the re-runs is the runner's limit, not the test's failure
            throw Error(MADSIM_E_LIMITS, "seed " + std::to_string(seed + i) + ": " + SimulationFailure::verdict_name(out[i].verdict) + " (a runner verdict, not a test failure) persists after re-runs with larger limits");
        }
        return out;
    }
};

}  // namespace runtime
}  // namespace madsim

#endif
