const char afx_build_id_str[] = "9746b05cf6b1";
