const char afx_build_id_str[] = "9e21f4fc5730";
